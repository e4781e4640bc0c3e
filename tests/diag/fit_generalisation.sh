#!/bin/bash
# Variant builds for tests/diag/fit_generalisation.py (CPU container: hipcc cross-compiles gfx950).  The constants come from
# tools/fit_backbone_terms.py --flags; the libraries travel to the GPU box with the snapshot (git-ignored *.so).
set -e
cd "$(dirname "$0")/../.."
C=trrosettax2-dynamics_amd/csrc
NMR=conf_2_1,conf_2_2,conf_1_3,conf_1_4
XRAY=conf_1_1,conf_1_2,conf_2_3,conf_2_4
make -C $C variant VARIANT=fitNMR  VFLAGS="$(python tools/fit_backbone_terms.py . --quiet --flags --decoys=$NMR)"
make -C $C variant VARIANT=fitXray VFLAGS="$(python tools/fit_backbone_terms.py . --quiet --flags --decoys=$XRAY)"
make -C $C variant VARIANT=fitoff  VFLAGS="-DTRX2_RAMA_FIT_ON=0 -DTRX2_OMEGA_FIT_ON=0"
make -C $C variant VARIANT=fitaa   VFLAGS="$(python tools/fit_backbone_terms.py . --quiet --flags --per-aa)"
