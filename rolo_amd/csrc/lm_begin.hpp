// The LM state at the start of a stage / frame: shared by passes.hip (rot_begin / frame_begin kernels, batches) and voxelmap.hip, whose
// finalize kernel — the last launch before a frame's LM chain — writes it on the way (one launch less per frame).
#pragma once
#include "rolo_internal.hpp"

namespace rolo {

// S = R R^T (six unique entries) — NOT assumed to be the identity: a caller's guess is a float matrix, orthonormal to 1e-7 only
__device__ inline void lm_set_rrt(double* S, const double* R) {
  S[0] = R[0] * R[0] + R[1] * R[1] + R[2] * R[2]; S[1] = R[0] * R[3] + R[1] * R[4] + R[2] * R[5]; S[2] = R[0] * R[6] + R[1] * R[7] + R[2] * R[8];
  S[3] = R[3] * R[3] + R[4] * R[4] + R[5] * R[5]; S[4] = R[3] * R[6] + R[4] * R[7] + R[5] * R[8]; S[5] = R[6] * R[6] + R[7] * R[7] + R[8] * R[8];
}

__device__ inline void rot_begin_dev(LmState* st, const RotBegin& a) {
  for (int i = 0; i < 9; i++) { st->xt_R[i] = a.R[i]; st->x0_R[i] = a.R[i]; st->tr_R[i] = a.R[i]; }
  lm_set_rrt(st->xt_S, st->xt_R);
  for (int i = 0; i < 6; i++) { st->x0_S[i] = st->xt_S[i]; st->tr_S[i] = st->xt_S[i]; }
  for (int i = 0; i < 3; i++) { st->xt_t[i] = a.t[i]; st->x0_t[i] = a.t[i]; }
  for (int i = 0; i < 36; i++) { st->H[i] = 0; st->final_H[i] = (i % 7 == 0) ? 1.0 : 0.0; }
  for (int i = 0; i < 6; i++) { st->b[i] = 0; st->d[i] = 0; }
  for (int i = 0; i < 9; i++) st->delta_R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; i++) st->delta_t[i] = 0;
  st->y0 = 0; st->lambda = -1.0; st->nu = 2.0;
  st->stage = 1; st->phase = 0; st->outer = 0; st->trial = 0; st->cur = 0; st->tr_cur = 0; st->n_corr = 0; st->tr_n_corr = 0;
  st->run_trans = a.run_trans;
  st->rot_done = 0; st->rot_converged = 0; st->rot_failed = 0; st->rot_outer = 0; st->rot_passes = 0; st->rot_ncorr = 0;
  st->trans_done = 0; st->trans_failed = 0; st->trans_outer = 0; st->trans_passes = 0;
  st->trace_count = 0; st->error = 0; st->pending = 0; st->lin_skip = 0; st->spec_lin = a.spec_lin; st->rot_cost_only = 0; st->trans_cost_only = 0; st->lmp_bailed = 0;
  st->optimizer = a.optimizer; st->max_iterations = a.max_iterations; st->fixed_iterations = a.fixed_iterations;
  st->lm_max = a.lm_max; st->q2_intended = a.q2_intended; st->rot_eps = a.rot_eps; st->trans_eps = a.trans_eps; st->lm_init = a.lm_init;
  st->inv_rot_eps = 1.0 / a.rot_eps; st->inv_trans_eps = 1.0 / a.trans_eps;
}


// rolo_register_async: both stages' inputs from the frame's argument block (pinned host memory or device)
__device__ inline void frame_begin_dev(LmState* st, const FrameArgs* a) {
  const RotBegin r = a->rot;
  const TransBegin t = a->trans;
  rot_begin_dev(st, r);
  for (int i = 0; i < 3; i++) { st->t0[i] = t.t0[i]; st->g[i] = t.g[i]; st->l[i] = t.l[i]; }
  st->dtn = t.dtn; st->dtn1 = t.dtn1; st->ct_lambda = t.ct_lambda;
}

}  // namespace rolo
