/* Plain-C caller of librdamd.so: the sequence RootDigger's model_t performs on a
 * corax_partition_t (src/model.cpp:159-168 create, :310 tips, :185/:337 model,
 * :367 P-matrices, :402 CLVs, :406 root lnL), written against
 * include/root_digger_amd.h only -- no Python, no C++ runtime, no HIP header.
 * Four taxa, one site "A" everywhere, Jukes-Cantor: the closed form is known
 * (tests/golden/single_jc.json holds the same case).  Prints the lnL; exit
 * status 0 when it matches the closed form to 1e-12. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "root_digger_amd.h"

static double jc_same(double t) { return 0.25 + 0.75 * exp(-4.0 * t / 3.0); }
static double jc_diff(double t) { return 0.25 - 0.25 * exp(-4.0 * t / 3.0); }

int main(void) {
  if (rdamd_device_count() < 1) {
    fprintf(stderr, "no HIP device: %s\n", rdamd_errmsg());
    return 2;
  }
  /* ((a:0.1,b:0.2):0.05,(c:0.3,d:0.4):0.05); rooted in the middle of the inner branch */
  const unsigned tips = 4, inner = 3;
  rdamd_partition_t *p = rdamd_partition_create(tips, inner, 4, 1, 1, 6, 1, inner, RDAMD_ATTRIB_NONREV);
  if (!p) { fprintf(stderr, "create: %s\n", rdamd_errmsg()); return 2; }
  for (unsigned t = 0; t < tips; ++t)
    if (rdamd_set_tip_states(p, t, rdamd_map_nt, "A") != RDAMD_SUCCESS) return 3;
  const double rates[12] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, freqs[4] = {.25, .25, .25, .25};
  const double one = 1.0;
  const unsigned w = 1, params_indices[1] = {0};
  rdamd_set_subst_params(p, 0, rates);
  rdamd_set_frequencies(p, 0, freqs);
  rdamd_set_category_rates(p, &one);
  rdamd_set_category_weights(p, &one);
  rdamd_set_pattern_weights(p, &w);

  const unsigned matrices[6] = {0, 1, 2, 3, 4, 5};
  const double lengths[6] = {0.1, 0.2, 0.3, 0.4, 0.05, 0.05};
  if (rdamd_update_prob_matrices(p, params_indices, matrices, lengths, 6) != RDAMD_SUCCESS) {
    fprintf(stderr, "update_prob_matrices: %s\n", rdamd_errmsg());
    return 3;
  }
  /* corax_operation_t field order: parent clv, parent scaler, child1 clv/matrix/scaler, child2 ... */
  const rdamd_operation_t ops[3] = {
      {4, 0, 0, 0, -1, 1, 1, -1},     /* (a,b)   */
      {5, 1, 2, 2, -1, 3, 3, -1},     /* (c,d)   */
      {6, 2, 4, 4, 0, 5, 5, 1},       /* root    */
  };
  rdamd_update_clvs(p, ops, 3);
  if (rdamd_errno()) { fprintf(stderr, "update_clvs: %s\n", rdamd_errmsg()); return 3; }
  const double lnl = rdamd_compute_root_loglikelihood(p, 6, 2, params_indices, NULL);

  /* closed form: sum over root state r and inner states x, y */
  double like = 0.0;
  for (int r = 0; r < 4; ++r)
    for (int x = 0; x < 4; ++x)
      for (int y = 0; y < 4; ++y) {
        const double prx = r == x ? jc_same(0.05) : jc_diff(0.05);
        const double pry = r == y ? jc_same(0.05) : jc_diff(0.05);
        const double ab = (x == 0 ? jc_same(0.1) : jc_diff(0.1)) * (x == 0 ? jc_same(0.2) : jc_diff(0.2));
        const double cd = (y == 0 ? jc_same(0.3) : jc_diff(0.3)) * (y == 0 ? jc_same(0.4) : jc_diff(0.4));
        like += 0.25 * prx * pry * ab * cd;
      }
  const double want = log(like);
  printf("lnL %.15f closed form %.15f\n", lnl, want);

  /* The same evaluation as one job of the fused evaluator that leaves only the root's two children
   * behind (rdamd_evaluate_root_children: what exhaustive_search needs between optimize_params
   * and its root-only steps, src/model.cpp:1154-1229), then the root-only evaluation the
   * reference's compute_lh_root makes on them (:415-446) at the same position. */
  double lnl2 = 0.0, lnl3 = 0.0;
  if (rdamd_evaluate_root_children(p, ops, 3, matrices, lengths, 6, rates, freqs, &one, &one, &lnl2) != RDAMD_SUCCESS) {
    fprintf(stderr, "evaluate_root_children: %s\n", rdamd_errmsg());
    return 3;
  }
  if (rdamd_root_loglikelihood_fused(p, &ops[2], params_indices, &lengths[4], &lengths[5], 1, &lnl3) != RDAMD_SUCCESS) {
    fprintf(stderr, "root_loglikelihood_fused: %s\n", rdamd_errmsg());
    return 3;
  }
  printf("root children %.15f root-only on them %.15f\n", lnl2, lnl3);
  rdamd_partition_destroy(p);
  return fabs(lnl - want) <= 1e-12 * fabs(want) && fabs(lnl2 - want) <= 1e-12 * fabs(want) &&
                 fabs(lnl3 - want) <= 1e-12 * fabs(want)
             ? 0
             : 1;
}
