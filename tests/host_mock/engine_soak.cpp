// engine_soak.cpp -- TEST ONLY: csrc/launch_engine.h compiled against mock_hip.h; T host threads hand single-decoy "folds" to the launch
// engines, sleep, are woken, FREE their buffers at once and come back with the next fold -- joining and leaving at different chunk
// boundaries, with launch classes mixed, some folds giving up at their launch cap, the engines' argument arrays filling up.
// Built and run by tests/test_host_sanitizers.py under -fsanitize=thread and -fsanitize=address.
#include "mock_hip.h"
#include "../../trrosettax2-dynamics_amd/csrc/launch_engine.h"
#include <unistd.h>

static std::atomic<long> g_done{0}, g_capped{0}, g_full{0};

static void fold_thread(int tid, int n_jobs) {
  unsigned rng = 12345u + 977u * (unsigned)tid;
  auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return (rng >> 8) & 0xffff; };
  hipStream_t own = pool_acquire(0);
  for (int j = 0; j < n_jobs; j++) {
    EngineJob job;
    int* evals = new int(0);
    int* done = new int(0);
    const bool capped = rnd() % 11 == 0;                     // a fold that never reports: given up at its cap
    int* pairs = new int(0);
    job.pa.evals = evals; job.pa.pairs = pairs; job.ca.pairs = pairs;
    job.ca.evals = evals; job.ca.done_count = done; job.ca.B = 1; job.ca.need = capped ? 1 << 30 : ENGINE_CHUNK * (1 + (int)(rnd() % 6)) - (int)(rnd() % ENGINE_CHUNK);
    job.done_count = done; job.B = 1; job.n_items = 1 + (int)(rnd() % 40); job.dyn = 1024 * (rnd() % 8);
    job.cls = (int)(rnd() % 3); job.fam_all = (int)(rnd() % 2); job.wave1 = (int)(rnd() % 2); job.segc = (int)(rnd() % 2);
    job.lowreg = (job.cls < 2 && rnd() % 2) ? 1 : 0;
    job.half = (job.lowreg && job.wave1 && job.segc && rnd() % 4 != 0) ? 1 : 0;   // half-evaluation launches (most of the folds that qualify)
    job.cap = capped ? 3 * ENGINE_CHUNK : 100000;
    hipEventCreate(&job.ready);
    hipEventRecord(job.ready, own);                          // the fold's start-up work on its own stream
    int rc;
    for (;;) {
      LaunchEngine* E = engine_pick(0);
      if (!E) { g_full++; std::this_thread::yield(); continue; }   // every engine full: the real fold would launch for itself
      rc = engine_run(E, &job);
      if (rc != 2) break;
      g_full++;
    }
    if (rc != 0 || job.state != 3) { fprintf(stderr, "fold failed: rc %d state %d err %s\n", rc, job.state, job.err.c_str()); _exit(2); }
    if (capped) { if (job.launches < job.cap) { fprintf(stderr, "capped fold released after %ld launches\n", job.launches); _exit(3); } g_capped++; }
    else if (*done != 1 || *evals < job.ca.need || job.done < 1) { fprintf(stderr, "fold woken before its decoy reported: done %d evals %d need %d\n", *done, *evals, job.ca.need); _exit(4); }
    // woken: by the engine's promise no launch in flight names these buffers any more -- give them back at once
    delete evals; delete done; delete pairs;
    hipEventDestroy(job.ready);
    g_done++;
    if (rnd() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rnd() % 300));
  }
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 16, J = argc > 2 ? atoi(argv[2]) : 8;
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back(fold_thread, t, J);
  for (auto& t : th) t.join();
  double st[2] = {0, 0};
  { std::lock_guard<std::mutex> lk(g_engine_mutex); for (LaunchEngine* E : g_engines[0]) { std::lock_guard<std::mutex> l2(E->mu); st[0] += E->st_chunks; st[1] += E->st_done; } }
  printf("engine soak ok: %ld folds on %d threads (%ld gave up at their cap, %ld found the engines full), %.0f chunks, %ld mock launches\n", g_done.load(), T, g_capped.load(),
         g_full.load(), st[0], g_mock_launches.load());
  if (g_done.load() != (long)T * J || st[1] != (double)(T * J)) return 5;
  fflush(stdout);
  _exit(0);     // the engines' and the mock streams' threads are detached and never end (as in the library): leave without unwinding under them
}
