/* Plain-C consumer of the batched entry points of include/dnlp_hip.h (tests/test_c_consumer.py): a stream of parametrised
 * batches (the serial loop of cvxpy/problems/problem.py:1256-1269) solved (1) one launch at a time with
 * dnlp_solve_batch_theta and (2) through dnlp_batch_stream_* with `slots` launches in flight — the two must agree bit for
 * bit.  usage: batch_stream tape.blob map.bin thetas.bin device batches batch slots
 *   map.bin:    int64 stride, int64 P, d0[stride], theta0[P], int64 indptr[stride + 1], int32 indices[nnz], vals[nnz]
 *   thetas.bin: batches x batch x P doubles
 * prints: seconds_one_at_a_time seconds_stream identical(0/1) optimal_instances */
#define _POSIX_C_SOURCE 199309L        /* clock_gettime under -std=c99 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "dnlp_hip.h"

static void* slurp(const char* path, long* len) {
  FILE* fp = fopen(path, "rb");
  if (!fp) { perror(path); exit(66); }
  fseek(fp, 0, SEEK_END);
  *len = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  void* buf = malloc((size_t)*len + 8);
  if (fread(buf, 1, (size_t)*len, fp) != (size_t)*len) { fprintf(stderr, "short read: %s\n", path); exit(66); }
  fclose(fp);
  return buf;
}
static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: %s tape.blob map.bin thetas.bin device batches batch slots\n", argv[0]); return 64; }
  long blen = 0, mlen = 0, tlen = 0;
  void* blob = slurp(argv[1], &blen);
  char* map = (char*)slurp(argv[2], &mlen);
  double* thetas = (double*)slurp(argv[3], &tlen);
  const int device = atoi(argv[4]), nb = atoi(argv[5]), B = atoi(argv[6]), slots = atoi(argv[7]);
  dnlp_problem* p = dnlp_create(blob, (size_t)blen, device);
  if (!p) { fprintf(stderr, "dnlp_create: %s\n", dnlp_last_error()); return 70; }
  /* what dnlp_amd.batch sets for every batch handle: a singular static pivot sequence is met with delta_c in the first
   * run, the Bunch-Kaufman switch only in the retry rungs */
  if (dnlp_set_option(p, "lazy_dense_fallback", "yes") != 0) { fprintf(stderr, "option: %s\n", dnlp_last_error()); return 70; }
  int64_t hdr[2];
  memcpy(hdr, map, 16);
  const int64_t stride = hdr[0], P = hdr[1];
  const double* d0 = (const double*)(map + 16);
  const double* theta0 = d0 + stride;
  const int64_t* indptr = (const int64_t*)(theta0 + P);
  const int64_t nnz = indptr[stride];
  const int32_t* indices = (const int32_t*)(indptr + stride + 1);
  /* (the values follow the int32 indices at the next 8-byte boundary) */
  const double* vals = (const double*)((const char*)indices + ((4 * (size_t)nnz + 7) & ~(size_t)7));
  if (dnlp_batch_stride(p) != stride) { fprintf(stderr, "stride %lld != %lld\n", (long long)dnlp_batch_stride(p), (long long)stride); return 70; }
  if (dnlp_batch_set_affine_map(p, (int)P, d0, theta0, indptr, indices, vals) != 0) { fprintf(stderr, "map: %s\n", dnlp_last_error()); return 70; }
  int64_t n64 = 0, m64 = 0;
  dnlp_dims(p, &n64, &m64, NULL, NULL);
  const size_t N = (size_t)n64, tot = (size_t)nb * (size_t)B;
  double* x1 = (double*)malloc(8 * tot * N); double* x2 = (double*)malloc(8 * tot * N);
  double* o1 = (double*)malloc(8 * tot); double* o2 = (double*)malloc(8 * tot);
  int* s1 = (int*)malloc(4 * tot); int* s2 = (int*)malloc(4 * tot);
  int* i1 = (int*)malloc(4 * tot); int* i2 = (int*)malloc(4 * tot);
  int* f1 = (int*)malloc(4 * tot); int* f2 = (int*)malloc(4 * tot);
  double sec = 0.0;
  /* warm both paths (first launch of a kernel form, buffer allocation) */
  if (dnlp_solve_batch_theta(p, B, thetas, (int)P, x1, o1, NULL, NULL, NULL, s1, i1, f1, &sec, NULL) != 0) { fprintf(stderr, "solve: %s\n", dnlp_last_error()); return 70; }
  dnlp_batch_stream* st = dnlp_batch_stream_create(p, slots);
  if (!st) { fprintf(stderr, "stream: %s\n", dnlp_last_error()); return 70; }
  for (int k = 0; k < slots; ++k) {
    const int t = dnlp_batch_stream_submit(st, B, thetas, (int)P, x2, o2, NULL, NULL, NULL, s2, i2, f2);
    if (t < 0 || dnlp_batch_stream_wait(st, t, NULL) != 0) { fprintf(stderr, "stream warm-up: %s\n", dnlp_last_error()); return 70; }
  }
  double t0 = now();
  for (int b = 0; b < nb; ++b) {
    const size_t o = (size_t)b * (size_t)B;
    if (dnlp_solve_batch_theta(p, B, thetas + o * (size_t)P, (int)P, x1 + o * N, o1 + o, NULL, NULL, NULL, s1 + o, i1 + o, f1 + o, &sec, NULL) != 0) {
      fprintf(stderr, "solve: %s\n", dnlp_last_error()); return 70;
    }
  }
  const double t_serial = now() - t0;
  t0 = now();
  int* tickets = (int*)malloc(sizeof(int) * (size_t)nb);
  for (int b = 0; b < nb; ++b) {
    const size_t o = (size_t)b * (size_t)B;
    if (b >= slots && dnlp_batch_stream_wait(st, tickets[b - slots], NULL) != 0) { fprintf(stderr, "wait: %s\n", dnlp_last_error()); return 70; }
    tickets[b] = dnlp_batch_stream_submit(st, B, thetas + o * (size_t)P, (int)P, x2 + o * N, o2 + o, NULL, NULL, NULL, s2 + o, i2 + o, f2 + o);
    if (tickets[b] < 0) { fprintf(stderr, "submit: %s\n", dnlp_last_error()); return 70; }
  }
  for (int b = nb > slots ? nb - slots : 0; b < nb; ++b)
    if (dnlp_batch_stream_wait(st, tickets[b], NULL) != 0) { fprintf(stderr, "wait: %s\n", dnlp_last_error()); return 70; }
  const double t_stream = now() - t0;
  dnlp_batch_stream_destroy(st);
  const int same = memcmp(x1, x2, 8 * tot * N) == 0 && memcmp(o1, o2, 8 * tot) == 0 && memcmp(s1, s2, 4 * tot) == 0 &&
                   memcmp(i1, i2, 4 * tot) == 0 && memcmp(f1, f2, 4 * tot) == 0;
  long optimal = 0;
  for (size_t k = 0; k < tot; ++k) optimal += s1[k] == 0;
  printf("%.6f %.6f %d %ld\n", t_serial, t_stream, same, optimal);
  dnlp_destroy(p);
  return same ? 0 : 1;
}
