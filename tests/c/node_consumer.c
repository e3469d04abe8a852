/* A plain C (C99) consumer of the multi-GPU host object of include/sipnet_amd.h (sipnet_node_*): the
 * ensemble sharded over the listed devices of one node, ONE RCCL all-gather of the statistics block
 * and ONE of the member-resolved planes, all from C -- no Python, no torch.  With one device the
 * collectives run in a one-rank communicator (what a one-GPU box can execute of the path).
 * usage: node_consumer <param file> <clim file> <n_members> <dev>[,<dev>...]
 *   -> key=value lines; exit 0 on success.  Without a GPU: create=100. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sipnet_amd.h"

int main(int argc, char **argv) {
  if (argc < 5) return 2;
  int32_t flags[SIPNET_NFLAGS] = {1, 1, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0}; /* context.c:35-53 */
  double raw[SIPNET_NPARAMS];
  int rc = sipnet_io_read_params(argv[1], flags, raw, NULL);
  if (rc) { printf("read_params=%d %s\n", rc, sipnet_last_error()); return 1; }
  sipnet_clim_table *clim = NULL;
  rc = sipnet_io_read_clim(argv[2], flags[SIPNET_F_GDD], &clim);
  if (rc) { printf("read_clim=%d %s\n", rc, sipnet_last_error()); return 1; }
  const int32_t T = sipnet_clim_nsteps(clim);
  const int32_t M = (int32_t)atoi(argv[3]);
  int32_t devices[64], nDev = 0;
  for (char *tok = strtok(argv[4], ","); tok && nDev < 64; tok = strtok(NULL, ",")) devices[nDev++] = (int32_t)atoi(tok);

  sipnet_node *nd = NULL;
  rc = sipnet_node_create(flags, 1, M, SIPNET_F64, devices, nDev, &nd);
  printf("create=%d\n", rc);
  if (rc == SIPNET_ERR_NO_DEVICE) {
    printf("no_device_message=%s\n", sipnet_last_error());
    sipnet_clim_free(clim);
    return sipnet_device_count() == 0 ? 0 : 1;
  }
  if (rc) { printf("error=%s\n", sipnet_last_error()); return 1; }
  printf("collective_library=%s\nn_devices=%d\n", sipnet_node_collective_library(nd), (int)sipnet_node_n_devices(nd));

  /* member m: aMax scaled by 1 + 0.001 m */
  double *members = (double *)malloc(sizeof(double) * (size_t)M * SIPNET_NPARAMS);
  const int iAmax = sipnet_param_index("aMax");
  for (int32_t m = 0; m < M; m++) {
    memcpy(members + (size_t)m * SIPNET_NPARAMS, raw, sizeof raw);
    members[(size_t)m * SIPNET_NPARAMS + iAmax] *= 1.0 + 0.001 * m;
  }
  rc = sipnet_node_set_math(nd, SIPNET_MATH_FAST);
  if (!rc) rc = sipnet_node_set_climate(nd, 0, T, sipnet_clim_data(clim), sipnet_clim_year(clim), sipnet_clim_day(clim));
  if (!rc) rc = sipnet_node_set_params(nd, 0, 0, M, members);
  if (!rc) rc = sipnet_node_setup(nd);
  if (!rc) rc = sipnet_node_run(nd, 0, T);
  double *total = (double *)malloc(sizeof(double) * 3 * (size_t)T * 2);
  if (!rc) rc = sipnet_node_gather_stats(nd, total);   /* ONE all-gather of the statistics block */
  if (!rc) rc = sipnet_node_gather_planes(nd);         /* ONE all-gather of the member-resolved planes */
  if (!rc) rc = sipnet_node_sync(nd);
  if (rc) { printf("run=%d %s\n", rc, sipnet_last_error()); return 1; }

  /* check on the host: the gathered planes (device 0's copy of everybody's) summed over all members
   * must give the statistics; every device's copy of the gathered statistics must be the same */
  const int64_t ld = sipnet_node_ld(nd);
  const size_t perDev = 3 * (size_t)T * (size_t)ld;
  double *g = (double *)malloc(sizeof(double) * perDev * (size_t)nDev);
  rc = sipnet_dev_to_host(g, sipnet_node_gathered_planes(nd, nDev - 1), sizeof(double) * perDev * (size_t)nDev, NULL);
  if (rc) { printf("copy=%d %s\n", rc, sipnet_last_error()); return 1; }
  double worst = 0.0, yearNee = 0.0, yearNee0 = 0.0, yearNeeLast = 0.0;
  for (int v = 0; v < 3; v++) {
    for (int32_t t = 0; t < T; t++) {
      double s1 = 0.0, s2 = 0.0;
      for (int32_t k = 0; k < nDev; k++) {
        int32_t first, count;
        sipnet_node_member_range(nd, k, &first, &count);
        const double *row = g + ((size_t)k * 3 + (size_t)v) * (size_t)T * (size_t)ld + (size_t)t * (size_t)ld;
        for (int32_t m = 0; m < count; m++) { s1 += row[m]; s2 += row[m] * row[m]; }
        if (v == 0 && k == 0) yearNee0 += row[0];
        if (v == 0 && k == nDev - 1) yearNeeLast += row[count - 1];
      }
      const double *st = total + ((size_t)v * (size_t)T + (size_t)t) * 2;
      double d1 = st[0] - s1, d2 = st[1] - s2;
      if (d1 < 0) d1 = -d1;
      if (d2 < 0) d2 = -d2;
      if (d1 > worst) worst = d1;
      if (d2 > worst) worst = d2;
      if (v == 0) yearNee += st[0];
    }
  }
  size_t statBlock = 3 * (size_t)T * 2;
  double *gs0 = (double *)malloc(sizeof(double) * statBlock * (size_t)nDev);
  double *gsk = (double *)malloc(sizeof(double) * statBlock * (size_t)nDev);
  sipnet_dev_to_host(gs0, sipnet_node_gathered_stats(nd, 0), sizeof(double) * statBlock * (size_t)nDev, NULL);
  sipnet_dev_to_host(gsk, sipnet_node_gathered_stats(nd, nDev - 1), sizeof(double) * statBlock * (size_t)nDev, NULL);
  printf("gathered_stats_identical=%d\n", memcmp(gs0, gsk, sizeof(double) * statBlock * (size_t)nDev) == 0);
  printf("stats_vs_planes_max_abs=%.3e\nsum_nee_ensemble=%.10f\nsum_nee_member_0=%.10f\nsum_nee_member_last=%.10f\n",
         worst, yearNee, yearNee0, yearNeeLast);
  printf("kernel=%s\n", sipnet_batch_last_kernel_name(sipnet_node_batch(nd, 0)));

  /* the same exchange overlapped with the computation: the run in 3 segments, each all-gathered on the shards'
   * second streams under the next segment's kernel -- the last device's copy must hold the planes gathered above */
  rc = sipnet_node_setup(nd);
  if (!rc) rc = sipnet_node_run_gathering(nd, 0, T, 3);
  if (!rc) rc = sipnet_node_sync(nd);
  if (rc) { printf("run_gathering=%d %s\n", rc, sipnet_last_error()); return 1; }
  int segmentsEqual = sipnet_node_n_segments(nd) == 3, covered = 0;
  for (int32_t j = 0; j < sipnet_node_n_segments(nd); j++) {
    int32_t first = -1, len = 0;
    void *dev = sipnet_node_gathered_segment(nd, nDev - 1, j, &first, &len);
    if (!dev || first != covered || len <= 0) { segmentsEqual = 0; break; }
    double *seg = (double *)malloc(sizeof(double) * (size_t)nDev * 3 * (size_t)len * (size_t)ld);
    rc = sipnet_dev_to_host(seg, dev, sizeof(double) * (size_t)nDev * 3 * (size_t)len * (size_t)ld, NULL);
    if (rc) { printf("copy_segment=%d %s\n", rc, sipnet_last_error()); return 1; }
    for (int32_t k = 0; k < nDev; k++)          /* segment layout [n_devices][3][len][ld] against [n_devices][3][T][ld] */
      for (int v = 0; v < 3; v++)
        if (memcmp(seg + (((size_t)k * 3 + (size_t)v) * (size_t)len) * (size_t)ld,
                   g + (((size_t)k * 3 + (size_t)v) * (size_t)T + (size_t)first) * (size_t)ld, sizeof(double) * (size_t)len * (size_t)ld) != 0)
          segmentsEqual = 0;
    free(seg);
    covered += len;
  }
  printf("gathering_segments_equal_gathered_planes=%d\n", segmentsEqual && covered == T);

  /* the member-resolved exchange in the form that fits under the kernel: every member's sums over groups of 48 steps (the
   * daily sums of a half-hourly forcing), reduced on the second stream and all-gathered per segment -- against the planes
   * gathered above, summed here in the same step order */
  {
    const int32_t K = 48, groups = (T + K - 1) / K;
    rc = sipnet_node_setup(nd);
    if (!rc) rc = sipnet_node_run_gathering_reduced(nd, 0, T, 4, SIPNET_GATHER_SUMS, K);
    if (!rc) rc = sipnet_node_sync(nd);
    if (rc) { printf("run_gathering_reduced=%d %s\n", rc, sipnet_last_error()); return 1; }
    int sumsEqual = sipnet_node_n_segments(nd) == 4, rowsSeen = 0;
    double sumNeeDay0Member0 = 0.0, worstSum = 0.0;
    size_t bytesPerRank = 0;
    for (int32_t j = 0; j < sipnet_node_n_segments(nd) && sumsEqual; j++) {
      int32_t first = -1, rows = 0, eb = 0;
      void *dev = sipnet_node_gathered_reduced(nd, nDev - 1, j, &first, &rows, &eb);
      if (!dev || first != rowsSeen || rows <= 0 || eb != 8) { sumsEqual = 0; break; }
      const size_t cnt = (size_t)nDev * 3 * (size_t)rows * (size_t)ld;
      double *seg = (double *)malloc(sizeof(double) * cnt);
      rc = sipnet_dev_to_host(seg, dev, sizeof(double) * cnt, NULL);
      if (rc) { printf("copy_reduced=%d %s\n", rc, sipnet_last_error()); return 1; }
      bytesPerRank += sizeof(double) * 3 * (size_t)rows * (size_t)ld;
      for (int32_t k = 0; k < nDev; k++)
        for (int v = 0; v < 3; v++)
          for (int32_t r = 0; r < rows; r++) {
            const int32_t gi = first + r, t0 = gi * K, t1 = (gi + 1) * K < T ? (gi + 1) * K : T;
            int32_t firstM, count;
            sipnet_node_member_range(nd, k, &firstM, &count);
            for (int32_t m = 0; m < count; m++) {
              double acc = 0.0;
              for (int32_t t = t0; t < t1; t++) acc += g[(((size_t)k * 3 + (size_t)v) * (size_t)T + (size_t)t) * (size_t)ld + (size_t)m];
              double d = seg[(((size_t)k * 3 + (size_t)v) * (size_t)rows + (size_t)r) * (size_t)ld + (size_t)m] - acc;
              if (d < 0) d = -d;
              if (d > worstSum) worstSum = d;
              if (k == 0 && v == 0 && gi == 0 && m == 0) sumNeeDay0Member0 = acc;
            }
          }
      free(seg);
      rowsSeen += rows;
    }
    printf("reduced_sums_equal_planes=%d\nreduced_sums_max_abs=%.3e\nreduced_groups=%d\nreduced_bytes_per_rank=%zu\nsum_nee_day0_member_0=%.12f\n",
           sumsEqual && rowsSeen == groups && worstSum == 0.0, worstSum, (int)groups, bytesPerRank, sumNeeDay0Member0);
  }
  free(g); free(gs0); free(gsk); free(total); free(members);
  sipnet_node_destroy(nd);
  sipnet_clim_free(clim);
  return 0;
}
