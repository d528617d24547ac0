// CPU test of rolo_amd/csrc/load_learner.hpp (the load a process cannot count, learned from its own frames' device time): synthetic frame durations, no GPU.
//   g++ -std=c++17 -I rolo_amd/csrc tests/cpp/learner_test.cpp -o learner_test && ./learner_test      (prints one line per scenario; exit code 0 = all held)
#include <cstdio>
#include "load_learner.hpp"

using rolo::LoadLearner;

static int fails = 0;
#define CHECK(c, what) do { if (!(c)) { std::printf("FAILED: %s (%s)\n", what, #c); fails++; } } while (0)

// frame time of a context as a function of the kernels it runs and of who else is on the chip (the measured numbers of profiles/r06/load_regimes.json, in ms)
static double frame_ms(bool busy_kernels, int regime) {
  switch (regime) {
    case 0: return busy_kernels ? 0.665 : 0.558;   // alone: 1 504 / 1 792 frames/s
    case 1: return busy_kernels ? 0.727 : 0.938;   // a second process with one context: 1 376 / 1 066
    default: return busy_kernels ? 0.680 : 0.582;  // beside a stream of 1 GiB copies: 1 471 / 1 719
  }
}

int main() {
  {  // alone: never leaves the idle-device kernels
    LoadLearner L; L.sizes(131072, 131072);
    for (int i = 0; i < 3000; i++) { L.frame(frame_ms(L.busy(), 0) * (1.0 + 0.02 * ((i * 7919) % 11 - 5) / 5.0)); CHECK(L.mode == 0, "alone: stays idle"); }
    std::printf("alone: mode %d best %.3f\n", L.mode, L.best_idle);
  }
  {  // a second process appears after 100 frames: tries the busy-device kernels, keeps them, looks again every 512 frames, and goes back when the other process leaves
    LoadLearner L; L.sizes(131072, 131072);
    int switched_at = -1, kept_at = -1, frames_busy = 0;
    for (int i = 0; i < 4000; i++) {
      const int regime = (i >= 100 && i < 3000) ? 1 : 0;
      const bool b = L.busy();
      frames_busy += b && regime == 1;
      L.frame(frame_ms(b, regime));
      if (switched_at < 0 && L.mode == 1) switched_at = i;
      if (kept_at < 0 && L.mode == 2) kept_at = i;
    }
    CHECK(switched_at >= 100 + LoadLearner::SETTLE && switched_at < 100 + 40, "shared: the try starts within 40 frames of the other process' arrival, after a settle period above the trigger");
    CHECK(kept_at > switched_at && kept_at <= switched_at + LoadLearner::SETTLE + 1, "shared: the busy-device kernels are kept after one settle period");
    CHECK(frames_busy > 0.9 * 2900, "shared: nine frames in ten of the shared period run the busy-device kernels");
    CHECK(L.mode == 0, "shared: back on the idle-device kernels after the other process has left");
    std::printf("shared: try at %d, kept at %d, %d of 2900 shared frames on the busy kernels, end mode %d\n", switched_at, kept_at, frames_busy, L.mode);
  }
  {  // best_idle learned alone, then a foreign copy stream: 4 % longer frames stay below the trigger
    LoadLearner L; L.sizes(131072, 131072);
    for (int i = 0; i < 2000; i++) { L.frame(frame_ms(L.busy(), i < 50 ? 0 : 2)); CHECK(L.mode == 0, "copies: stays idle"); }
    std::printf("copies: mode %d\n", L.mode);
  }
  {  // a load under which the busy-device kernels do NOT pay (frames 40 % longer either way): one try, back, and no second try for RECHECK frames
    LoadLearner L; L.sizes(65536, 65536);
    int tries = 0, prev = 0;
    for (int i = 0; i < 400; i++) { const double ms = (i < 50 ? 1.0 : 1.4) * (L.busy() ? 0.40 : 0.35); L.frame(ms); tries += L.mode == 1 && prev != 1; prev = L.mode; }
    CHECK(tries == 1 && L.mode == 0 && L.holdoff > 0, "no gain: one try, then the hold-off");
    std::printf("no gain: %d tries, mode %d, hold-off %d\n", tries, L.mode, L.holdoff);
  }
  {  // another cloud size is another workload
    LoadLearner L; L.sizes(1000, 1000); for (int i = 0; i < 20; i++) L.frame(0.1);
    L.sizes(2000, 1000);
    CHECK(L.best_idle == 0 && L.frames == 0 && L.n_src == 2000, "sizes: reset");
  }
  std::printf(fails ? "%d check(s) failed\n" : "all held\n", fails);
  return fails ? 1 : 0;
}
