// What the per-process count of frames in flight cannot see — another PROCESS (or a foreign workload) on the same GPU — learned from the device's own signal: how long a
// context's frames take on the device (round 5's verdict, item 7). Pure host logic, no HIP: api.hip feeds it the event-timed duration of every frame that was sized by it
// (rolo_set_load_hint -1, no other frame of this process in flight, not sharded); tests/cpp/learner_test.cpp drives it with synthetic durations on the CPU tier.
//
// While the context believes the device idle (mode 0) it remembers the shortest frame it has seen at the current sizes. Frames that last a quarter longer than that for a
// while mean somebody else is using the chip: the context TRIES the busy-device kernels (mode 1: 64-query packet walk, 64 resident LM workgroups) for a dozen frames. If its
// frames get shorter (below 0.95 of what they were when it switched) it KEEPS them (mode 2) and looks again after 512 frames — is the other user still there? — otherwise it
// goes back and does not try again for 512 frames. Measured (profiles/r06/load_regimes.json): two processes with one context each, 2 x 1 066 scans/s on the idle-device
// kernels, 2 x 1 376 on the busy-device ones — which both find by themselves within a second; beside a stream of 1 GiB copies the frames are 4 % longer, below the trigger,
// and the idle-device kernels (1 719 against 1 471 scans/s) stay.
#pragma once

namespace rolo {

struct LoadLearner {
  int mode = 0;              // 0: idle-device kernels, 1: trying the busy-device kernels, 2: keeping them
  int frames = 0;            // frames in this mode
  double best_idle = 0, ema = 0, idle_at_switch = 0;
  int n_src = 0, n_tgt = 0;  // the sizes best_idle belongs to
  int holdoff = 0;           // frames before the next try after one that did not pay
  int over = 0;              // consecutive frames whose running time was above the trigger

  static constexpr int SETTLE = 12;          // frames a mode is looked at before a decision (the first two of a mode are eager launches and a graph capture)
  static constexpr int RECHECK = 512;        // frames the busy-device kernels are kept before the idle-device ones are looked at again; also the hold-off after a try that did not pay
  static constexpr double TRIGGER = 1.25;    // the running frame time over the best one that starts a try
  static constexpr double KEEP = 0.95;       // the trial's frame time over the one it started from that makes it stay

  // the sizes the next frame has: another cloud size is another workload — start over
  void sizes(int ns, int nt) { if (ns != n_src || nt != n_tgt) { *this = LoadLearner{}; n_src = ns; n_tgt = nt; } }
  bool busy() const { return mode != 0; }

  // one finished frame of `ms` milliseconds on the device
  void frame(double ms) {
    if (!(ms > 0)) return;
    frames++;
    ema = frames <= 3 ? ms : 0.8 * ema + 0.2 * ms;   // (frames 1-2 of a mode: eager launches and the capture — the average restarts behind them)
    if (mode == 0) {
      if (frames >= 3 && (best_idle == 0 || ms < best_idle)) best_idle = ms;
      if (holdoff > 0) holdoff--;
      // the running average must stay above the trigger for a settle period before the try starts: it is what the try will be compared with, and right after the other
      // user arrives it has only begun to rise (a try started on the first frame above the trigger was judged against 0.70 ms where the shared frames last 0.94, lost,
      // and cost 512 frames of hold-off: tests/cpp/learner_test.cpp)
      over = (frames >= SETTLE && holdoff == 0 && best_idle > 0 && ema > TRIGGER * best_idle) ? over + 1 : 0;
      if (over >= SETTLE) { idle_at_switch = ema; mode = 1; frames = 0; over = 0; }
    } else if (mode == 1) {
      if (frames >= SETTLE) {
        if (ema < KEEP * idle_at_switch) { mode = 2; frames = 0; }
        else { mode = 0; frames = 0; holdoff = RECHECK; }
      }
    } else if (frames >= RECHECK) { mode = 0; frames = 0; holdoff = 0; }
  }
};

}  // namespace rolo
