/* A plain C (C99) host in which every GPU has a PROCESS of its own (an MPI-style layout; sipnet_node_* is the other one: one
 * process, several devices): rank r of `world` owns members [r M / world, (r + 1) M / world) of the ensemble on device `device`,
 * runs the year with every member's daily sums formed inside the step kernel's launch (sipnet_batch_run_sums) and all-gathers
 * that block through the engine's own RCCL communicator on the batch's stream (sipnet_comm_*).  The 128-byte communicator id
 * travels through a file rank 0 writes (a launcher's broadcast would do the same).
 * usage: rank_consumer <param file> <clim file> <n_members> <world> <rank> <device> <id file>
 *   -> key=value lines; exit 0 on success.  Without a GPU: create=100. */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sipnet_amd.h"

int main(int argc, char **argv) {
  if (argc < 8) return 2;
  int32_t flags[SIPNET_NFLAGS] = {1, 1, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0}; /* context.c:35-53 */
  double raw[SIPNET_NPARAMS];
  int rc = sipnet_io_read_params(argv[1], flags, raw, NULL);
  if (rc) { printf("read_params=%d %s\n", rc, sipnet_last_error()); return 1; }
  sipnet_clim_table *clim = NULL;
  rc = sipnet_io_read_clim(argv[2], flags[SIPNET_F_GDD], &clim);
  if (rc) { printf("read_clim=%d %s\n", rc, sipnet_last_error()); return 1; }
  const int32_t T = sipnet_clim_nsteps(clim), M = (int32_t)atoi(argv[3]);
  const int32_t world = (int32_t)atoi(argv[4]), rank = (int32_t)atoi(argv[5]), device = (int32_t)atoi(argv[6]);
  const int32_t first = (int32_t)((int64_t)M * rank / world), mine = (int32_t)((int64_t)M * (rank + 1) / world) - first;
  const int32_t most = (M + world - 1) / world;   /* every rank's block has the same size: the largest share */

  sipnet_batch *b = NULL;
  rc = sipnet_batch_create(flags, 1, mine, SIPNET_F64, device, &b);
  printf("create=%d\n", rc);
  if (rc == SIPNET_ERR_NO_DEVICE) {
    printf("no_device_message=%s\n", sipnet_last_error());
    sipnet_clim_free(clim);
    return sipnet_device_count() == 0 ? 0 : 1;
  }
  if (rc) { printf("error=%s\n", sipnet_last_error()); return 1; }

  /* the communicator: rank 0 makes the id, the others read it */
  uint8_t id[128];
  if (rank == 0) {
    rc = sipnet_comm_unique_id(id);
    if (rc) { printf("unique_id=%d %s\n", rc, sipnet_last_error()); return 1; }
    FILE *f = fopen(argv[7], "wb");
    if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 1;
    fclose(f);
  } else {
    FILE *f = NULL;
    for (int tries = 0; tries < 600 && !f; tries++) {
      f = fopen(argv[7], "rb");
      if (f && fread(id, 1, sizeof id, f) != sizeof id) { fclose(f); f = NULL; }
      if (!f) { struct timespec ts = {0, 100000000}; nanosleep(&ts, NULL); }
    }
    if (!f) { printf("no id file\n"); return 1; }
    fclose(f);
  }
  sipnet_comm *comm = NULL;
  rc = sipnet_comm_create(id, world, rank, device, &comm);
  if (rc) { printf("comm_create=%d %s\n", rc, sipnet_last_error()); return 1; }
  printf("comm_world=%d\n", (int)sipnet_comm_world(comm));

  /* member m of the whole ensemble: aMax scaled by 1 + 0.001 m */
  double *members = (double *)malloc(sizeof(double) * (size_t)mine * SIPNET_NPARAMS);
  const int iAmax = sipnet_param_index("aMax");
  for (int32_t m = 0; m < mine; m++) {
    memcpy(members + (size_t)m * SIPNET_NPARAMS, raw, sizeof raw);
    members[(size_t)m * SIPNET_NPARAMS + iAmax] *= 1.0 + 0.001 * (first + m);
  }
  const int32_t K = 48, groups = (T + K - 1) / K;
  const size_t block = (size_t)3 * groups * most;   /* doubles per rank: [3][groups][most] */
  double *d_mine = (double *)sipnet_dev_alloc(sizeof(double) * block);
  double *d_all = (double *)sipnet_dev_alloc(sizeof(double) * block * (size_t)world);
  if (!d_mine || !d_all) { printf("alloc failed\n"); return 1; }
  rc = sipnet_batch_set_math(b, SIPNET_MATH_FAST);
  if (!rc) rc = sipnet_batch_set_climate(b, 0, T, sipnet_clim_data(clim), sipnet_clim_year(clim), sipnet_clim_day(clim));
  if (!rc) rc = sipnet_batch_set_params(b, 0, 0, mine, members);
  if (!rc) rc = sipnet_batch_setup(b, NULL);
  printf("sums_in_kernel=%d\n", (int)sipnet_batch_sums_in_kernel(b));
  if (!rc) rc = sipnet_batch_run_sums(b, 0, T, K, d_mine, d_mine + (size_t)groups * most, d_mine + 2 * (size_t)groups * most, most, NULL);
  if (!rc) rc = sipnet_comm_all_gather(comm, d_mine, d_all, (int64_t)(sizeof(double) * block), NULL);   /* same (null) stream: ordered */
  double *all = (double *)malloc(sizeof(double) * block * (size_t)world);
  if (!rc) rc = sipnet_dev_to_host(all, d_all, sizeof(double) * block * (size_t)world, NULL);
  if (rc) { printf("run=%d %s\n", rc, sipnet_last_error()); return 1; }
  printf("kernel=%s\n", sipnet_batch_last_kernel_name(b));

  /* what this rank now holds of everybody: the year's NEE of the ensemble's first and last member, day 0 of member 0 */
  const int32_t lastRank = world - 1, lastCount = M - (int32_t)((int64_t)M * lastRank / world);
  double year0 = 0.0, yearLast = 0.0;
  for (int32_t g = 0; g < groups; g++) {
    year0 += all[(size_t)g * most];
    yearLast += all[(size_t)lastRank * block + (size_t)g * most + (size_t)(lastCount - 1)];
  }
  printf("groups=%d\nbytes_per_rank=%lld\nsum_nee_member_0=%.17g\nsum_nee_member_last=%.17g\nsum_nee_day0_member_0=%.17g\n", (int)groups,
         (long long)(sizeof(double) * block), year0, yearLast, all[0]);
  sipnet_comm_destroy(comm);
  sipnet_dev_free(d_mine);
  sipnet_dev_free(d_all);
  sipnet_batch_destroy(b);
  sipnet_clim_free(clim);
  free(members);
  free(all);
  printf("rc=0\n");
  return 0;
}
