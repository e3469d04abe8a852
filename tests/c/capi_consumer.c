/* A plain C (C99) consumer of include/sipnet_amd.h -- what a sipnet maintainer's frontend.c would
 * compile: checks that the header is valid C (no C++-isms), that the library links from C, and the
 * host-side entry points behave (file readers, parameter table, formatter, error reporting).
 * With a GPU it also runs a 2-member batch; without one it expects SIPNET_ERR_NO_DEVICE.
 * usage: capi_consumer <param file> <clim file>   -> prints key=value lines, exit 0 on success */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sipnet_amd.h"

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  int32_t flags[SIPNET_NFLAGS] = {1, 1, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0}; /* context.c:35-53 */
  double raw[SIPNET_NPARAMS];
  int rc = sipnet_io_read_params(argv[1], flags, raw, NULL);
  if (rc) { printf("read_params=%d %s\n", rc, sipnet_last_error()); return 1; }
  sipnet_clim_table *clim = NULL;
  rc = sipnet_io_read_clim(argv[2], flags[SIPNET_F_GDD], &clim);
  if (rc) { printf("read_clim=%d %s\n", rc, sipnet_last_error()); return 1; }
  const int32_t T = sipnet_clim_nsteps(clim);
  printf("version=%s\nn_steps=%d\naMax=%.6f\nindex_aMax=%d\n", sipnet_version(), (int)T,
         raw[sipnet_param_index("aMax")], (int)sipnet_param_index("aMax"));
  char buf[2048];
  int n = sipnet_io_format_out_header(buf, sizeof buf);
  printf("header_bytes=%d\n", n);

  const int32_t M = 2;
  sipnet_batch *b = NULL;
  rc = sipnet_batch_create(flags, 1, M, SIPNET_F64, 0, &b);
  printf("create=%d\n", rc);
  if (rc == SIPNET_ERR_NO_DEVICE) { /* no CPU fallback exists: this is the expected answer here */
    printf("no_device_message=%s\n", sipnet_last_error());
    sipnet_clim_free(clim);
    return sipnet_device_count() == 0 ? 0 : 1;
  }
  if (rc) return 1;
  double members[2 * SIPNET_NPARAMS];
  memcpy(members, raw, sizeof raw);
  memcpy(members + SIPNET_NPARAMS, raw, sizeof raw);
  members[SIPNET_NPARAMS + sipnet_param_index("aMax")] *= 1.1;
  rc = sipnet_batch_set_math(b, SIPNET_MATH_STRICT);
  /* (the several-sites entry with one site: its copies run on the batch's plan threads) */
  const int32_t n_steps[1] = {T};
  const double *cl[1] = {sipnet_clim_data(clim)};
  const int32_t *yr[1] = {sipnet_clim_year(clim)}, *dy[1] = {sipnet_clim_day(clim)};
  if (!rc) rc = sipnet_batch_set_climate_sites(b, 0, 1, n_steps, cl, yr, dy);
  if (!rc) rc = sipnet_batch_set_params(b, 0, 0, M, members);
  if (!rc) rc = sipnet_batch_setup(b, NULL);
  double *d_nee = (double *)sipnet_dev_alloc(sizeof(double) * (size_t)T * M);
  if (!rc && !d_nee) rc = 1;
  if (!rc) rc = sipnet_batch_run(b, 0, T, d_nee, NULL, NULL, NULL, M, NULL);
  double *nee = (double *)malloc(sizeof(double) * (size_t)T * M);
  if (!rc) rc = sipnet_dev_to_host(nee, d_nee, sizeof(double) * (size_t)T * M, NULL);
  if (rc) { printf("run=%d %s\n", rc, sipnet_last_error()); return 1; }
  double s0 = 0, s1 = 0;
  for (int32_t t = 0; t < T; t++) { s0 += nee[(size_t)t * M]; s1 += nee[(size_t)t * M + 1]; }
  printf("sum_nee_0=%.10f\nsum_nee_1=%.10f\n", s0, s1);
  free(nee);
  sipnet_dev_free(d_nee);
  sipnet_batch_destroy(b);
  sipnet_clim_free(clim);
  return 0;
}
