/* A plain C (C99) consumer of the particle filter's exchange step behind the C-ABI (BASELINE config 5;
 * include/sipnet_amd.h: sipnet_node_pf_connect / sipnet_node_forecast / sipnet_node_pf_analysis): K cycles of
 * setupModel -> one-day forecast -> analysis on a node of the listed devices -- ONE all-gather of the
 * log-weight blocks (RCCL with one device per shard), then every shard resamples by reading its ancestors where
 * they live -- timed against the same K cycles on ONE batch without any exchange path
 * (sipnet_batch_pf_analysis), with nothing synchronised inside either loop.  With one shard the two must end
 * in the same state, bit for bit -- and with several whenever the shards and the twin take the SAME step kernel: the
 * kernel follows the shape (two shards of 65 536 particles take the four-chunk cooperative kernel, the twin's 131 072
 * the one-wave kernel: equal to rounding, not to bits), so both kernels' names are printed next to state_identical and
 * the largest relative difference, and a seventh argument `1` puts node and twin on the one-wave kernel.
 * usage: pf_consumer <param file> <clim file> <n_particles> <dev>[,<dev>...] <cycles> <n_steps> [one_wave=0|1]
 *   -> key=value lines; exit 0 on success.  Without a GPU: create=100. */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sipnet_amd.h"

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  int32_t flags[SIPNET_NFLAGS] = {1, 1, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0}; /* context.c:35-53 */
  double raw[SIPNET_NPARAMS];
  int rc = sipnet_io_read_params(argv[1], flags, raw, NULL);
  if (rc) { printf("read_params=%d %s\n", rc, sipnet_last_error()); return 1; }
  sipnet_clim_table *clim = NULL;
  rc = sipnet_io_read_clim(argv[2], flags[SIPNET_F_GDD], &clim);
  if (rc) { printf("read_clim=%d %s\n", rc, sipnet_last_error()); return 1; }
  const int32_t N = (int32_t)atoi(argv[3]);
  int32_t devices[64], nDev = 0;
  for (char *tok = strtok(argv[4], ","); tok && nDev < 64; tok = strtok(NULL, ",")) devices[nDev++] = (int32_t)atoi(tok);
  const int K = atoi(argv[5]);
  const int32_t T = (int32_t)atoi(argv[6]);
  if (T > sipnet_clim_nsteps(clim)) return 2;
  const int oneWave = argc > 7 && atoi(argv[7]) != 0;

  sipnet_node *nd = NULL;
  rc = sipnet_node_create_sharded(flags, 1, N, SIPNET_F32_MIXED, devices, nDev, SIPNET_SHARD_MEMBERS, &nd);
  printf("create=%d\n", rc);
  if (rc == SIPNET_ERR_NO_DEVICE) {
    printf("no_device_message=%s\n", sipnet_last_error());
    sipnet_clim_free(clim);
    return sipnet_device_count() == 0 ? 0 : 1;
  }
  if (rc) { printf("error=%s\n", sipnet_last_error()); return 1; }
  printf("collective_library=%s\nn_shards=%d\n", sipnet_node_collective_library(nd), (int)sipnet_node_n_devices(nd));

  /* particle m: aMax and baseVegResp scaled by a deterministic pseudo-random factor */
  double *members = (double *)malloc(sizeof(double) * (size_t)N * SIPNET_NPARAMS);
  const int iAmax = sipnet_param_index("aMax"), iResp = sipnet_param_index("baseVegResp");
  uint32_t x = 12345u;
  for (int32_t m = 0; m < N; m++) {
    memcpy(members + (size_t)m * SIPNET_NPARAMS, raw, sizeof raw);
    x = x * 1664525u + 1013904223u;
    members[(size_t)m * SIPNET_NPARAMS + iAmax] *= 0.8 + 0.4 * (double)(x >> 8) / 16777216.0;
    x = x * 1664525u + 1013904223u;
    members[(size_t)m * SIPNET_NPARAMS + iResp] *= 0.8 + 0.4 * (double)(x >> 8) / 16777216.0;
  }
  /* the twin: ONE batch on the first device, no exchange path */
  sipnet_batch *b = NULL;
  rc = sipnet_batch_create(flags, 1, N, SIPNET_F32_MIXED, devices[0], &b);
  if (!rc && oneWave) rc = sipnet_batch_set_kernel(b, SIPNET_KERNEL_ONE_WAVE, 0);
  if (!rc && oneWave) rc = sipnet_node_set_kernel(nd, SIPNET_KERNEL_ONE_WAVE, 0);
  if (!rc) rc = sipnet_batch_set_climate(b, 0, T, sipnet_clim_data(clim), sipnet_clim_year(clim), sipnet_clim_day(clim));
  if (!rc) rc = sipnet_batch_set_params(b, 0, 0, N, members);
  if (!rc) rc = sipnet_node_set_climate(nd, 0, T, sipnet_clim_data(clim), sipnet_clim_year(clim), sipnet_clim_day(clim));
  if (!rc) rc = sipnet_node_set_params(nd, 0, 0, N, members);
  if (!rc) rc = sipnet_node_setup(nd);
  if (!rc) rc = sipnet_node_pf_connect(nd, 1);
  if (rc) { printf("setup=%d %s\n", rc, sipnet_last_error()); return 1; }

  float *planes = (float *)sipnet_dev_alloc(sizeof(float) * 3 * (size_t)T * (size_t)N);
  double *logw = (double *)sipnet_dev_alloc(sizeof(double) * (size_t)N);
  int32_t *anc = (int32_t *)sipnet_dev_alloc(sizeof(int32_t) * (size_t)N);
  int64_t *totals = (int64_t *)sipnet_dev_alloc(sizeof(int64_t) * 64);
  float *nee = planes, *gpp = planes + (size_t)T * N, *et = planes + 2 * (size_t)T * N;

  /* the observation: the first particle's own daily NEE (from a first forecast), sigma wide enough to keep many */
  rc = sipnet_batch_setup(b, NULL);
  if (!rc) rc = sipnet_batch_run(b, 0, T, nee, gpp, et, NULL, N, NULL);
  float *col = (float *)malloc(sizeof(float) * (size_t)T * (size_t)N);
  if (!rc) rc = sipnet_dev_to_host(col, nee, sizeof(float) * (size_t)T * (size_t)N, NULL);
  if (rc) { printf("first_forecast=%d %s\n", rc, sipnet_last_error()); return 1; }
  double obs = 0.0, lo = 1e300, hi = -1e300;
  for (int32_t m = 0; m < N; m += (N > 4096 ? N / 4096 : 1)) {
    double s = 0.0;
    for (int32_t t = 0; t < T; t++) s += col[(size_t)t * N + m];
    if (s < lo) lo = s;
    if (s > hi) hi = s;
    if (m == 0) obs = s;
  }
  free(col);
  const double sigma = (hi - lo) * 0.25 + 1e-9;
  obs = 0.5 * (lo + hi);

  double msNode = 0.0, msPlain = 0.0;
  for (int rep = 0; rep < 2; rep++) {   /* rep 0 warms both paths up */
    const int cycles = rep == 0 ? 3 : K;
    sipnet_node_sync(nd);
    double t0 = now_ms();
    for (int c = 0; c < cycles && !rc; c++) {
      rc = sipnet_node_setup(nd);
      if (!rc) rc = sipnet_node_pf_arm(nd, obs, sigma);   /* (the forecast's launch leaves the log-weight block itself) */
      if (!rc) rc = sipnet_node_forecast(nd, 0, T);
      if (!rc) rc = sipnet_node_pf_analysis(nd, 0, obs, sigma, 0.5);
    }
    if (!rc) rc = sipnet_node_sync(nd);
    msNode = (now_ms() - t0) / cycles;
    int32_t checked = 0;
    if (!rc) rc = sipnet_node_pf_check(nd, &checked);
    if (rc) { printf("node_cycles=%d %s\n", rc, sipnet_last_error()); return 1; }
    sipnet_stream_sync(NULL);
    t0 = now_ms();
    for (int c = 0; c < cycles && !rc; c++) {
      rc = sipnet_batch_setup(b, NULL);
      if (!rc) rc = sipnet_batch_pf_arm(b, obs, sigma, logw);
      if (!rc) rc = sipnet_batch_run(b, 0, T, nee, gpp, et, NULL, N, NULL);
      if (!rc) rc = sipnet_batch_pf_analysis(b, nee, 1, T, N, obs, sigma, 0.5, 1, logw, anc, totals + (c & 63), NULL);
    }
    if (!rc) rc = sipnet_stream_sync(NULL);
    msPlain = (now_ms() - t0) / cycles;
    if (rc) { printf("plain_cycles=%d %s\n", rc, sipnet_last_error()); return 1; }
  }
  printf("cycles=%d\nms_per_cycle_node=%.4f\nms_per_cycle_plain=%.4f\noverhead_ms=%.4f\n", K, msNode, msPlain, msNode - msPlain);
  printf("predicted_weak_scaling_efficiency=%.3f\n", msPlain / msNode);

  /* both paths ran the same K + 3 cycles from the same inputs: with ONE shard the states must agree bit for bit;
   * with several, shard k's particles are the twin's [first_k, first_k + count_k) */
  double *sNode = (double *)malloc(sizeof(double) * (size_t)N * SIPNET_NSTATE);
  double *sTwin = (double *)malloc(sizeof(double) * (size_t)N * SIPNET_NSTATE);
  rc = sipnet_batch_get_state(b, sTwin, NULL);
  for (int32_t k = 0; k < nDev && !rc; k++) {
    int32_t first, count;
    sipnet_node_member_range(nd, k, &first, &count);
    rc = sipnet_batch_get_state(sipnet_node_batch(nd, k), sNode + (size_t)first * SIPNET_NSTATE, sipnet_node_stream(nd, k));
  }
  if (rc) { printf("get_state=%d %s\n", rc, sipnet_last_error()); return 1; }
  printf("state_identical=%d\n", memcmp(sNode, sTwin, sizeof(double) * (size_t)N * SIPNET_NSTATE) == 0);
  {
    double worst = 0.0;   /* (pools only: the first 13 words of a particle's state) */
    for (int32_t m = 0; m < N; m++)
      for (int q = 0; q < 13; q++) {
        const double u = sNode[(size_t)m * SIPNET_NSTATE + q], v = sTwin[(size_t)m * SIPNET_NSTATE + q];
        double d = u - v, sc = (u < 0 ? -u : u) + (v < 0 ? -v : v) + 1e-30;
        if (d < 0) d = -d;
        if (d / sc > worst) worst = d / sc;
      }
    printf("state_max_rel_diff=%.3e\nkernel_twin=%s\n", worst, sipnet_batch_last_kernel_name(b));
  }
  {
    sipnet_pf_info info;   /* what the exchange says about itself */
    int64_t crossing = 0, cyclesSeen = 0;
    int32_t byIndex = 0, fused = 0, grid = 0;
    for (int32_t k = 0; k < nDev; k++)
      if (sipnet_batch_pf_info(sipnet_node_batch(nd, k), &info, sipnet_node_stream(nd, k)) == 0) {
        crossing += info.crossing;
        cyclesSeen = info.cycles;
        byIndex = info.params_by_index;
        fused = info.fused;
        grid = info.grid;
      }
    printf("crossing_per_cycle=%.1f\nparams_by_index=%d\nanalysis_one_launch=%d\nanalysis_grid=%d\n",
           cyclesSeen ? (double)crossing / (double)cyclesSeen : 0.0, (int)byIndex, (int)fused, (int)grid);
  }
  int64_t distinct = 0;
  for (int32_t m = 1; m < N; m++) distinct += sNode[(size_t)m * SIPNET_NSTATE] != sNode[(size_t)(m - 1) * SIPNET_NSTATE];
  printf("distinct_neighbours=%lld\n", (long long)distinct);
  printf("kernel=%s\n", sipnet_batch_last_kernel_name(sipnet_node_batch(nd, 0)));
  free(sNode); free(sTwin); free(members);
  sipnet_dev_free(planes); sipnet_dev_free(logw); sipnet_dev_free(anc); sipnet_dev_free(totals);
  sipnet_batch_destroy(b);
  sipnet_node_destroy(nd);
  sipnet_clim_free(clim);
  return 0;
}
