/* The mmap alternative to tools/shm_write_probe.c: a fresh 9 GB file in /dev/shm is sized, mapped MAP_SHARED and its
 * pages are brought in ahead of time (fallocate + MAP_POPULATE, or MAP_POPULATE alone); N threads then memcpy 32 MB
 * chunks from private buffers into the mapping -- user-space stores instead of pwrite's kernel copy.
 * gcc -O2 -pthread tools/shm_mmap_probe.c -o tools/_bin/shm_mmap_probe */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static const size_t CHUNK = 32u << 20;
static long long total;
static int nthreads;
static long long next_chunk;
static char *base;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;

static void *work(void *arg) {
  char *buf = malloc(CHUNK);
  memset(buf, 1 + (int)(long)arg, CHUNK);
  for (;;) {
    pthread_mutex_lock(&mu);
    long long c = next_chunk++;
    pthread_mutex_unlock(&mu);
    long long off = c * (long long)CHUNK;
    if (off >= total) break;
    size_t n = (size_t)(total - off < (long long)CHUNK ? total - off : (long long)CHUNK);
    memcpy(base + off, buf, n);
  }
  free(buf);
  return NULL;
}
static double pass(void) {
  pthread_t th[64];
  next_chunk = 0;
  double t0 = now();
  for (long t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, work, (void *)t);
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  return now() - t0;
}
int main(int argc, char **argv) {
  const char *path = argc > 1 ? argv[1] : "/dev/shm/shm_probe.bin";
  total = (argc > 2 ? atoll(argv[2]) : 9000ll) << 20;
  int counts[] = {1, 2, 4, 8};
  for (int mode = 0; mode < 3; ++mode)
    for (int i = 0; i < 4; ++i) {
      nthreads = counts[i];
      unlink(path);
      int fd = open(path, O_CREAT | O_RDWR, 0644);
      double t0 = now(), tf = 0, tm;
      int rc = 0;
      if (mode == 0) { rc = posix_fallocate(fd, 0, total); tf = now() - t0; }
      else rc = ftruncate(fd, total);
      double t1 = now();
      base = mmap(NULL, (size_t)total, PROT_READ | PROT_WRITE, MAP_SHARED | (mode < 2 ? MAP_POPULATE : 0), fd, 0);
      tm = now() - t1;
      if (base == MAP_FAILED) { perror("mmap"); return 1; }
      double a = pass(), b = pass();
      t1 = now();
      munmap(base, (size_t)total);
      close(fd);
      double tu = now() - t1;
      unlink(path);
      printf("%s, %d threads: size %.2f s (rc %d) + mmap %.2f s, memcpy into the mapping %.2f s (%.1f GB/s), again %.2f s (%.1f GB/s), munmap + close %.2f s\n",
             mode == 0 ? "fallocate + MAP_POPULATE" : mode == 1 ? "ftruncate + MAP_POPULATE" : "ftruncate, no populate  ", nthreads, tf, rc, tm, a,
             total / a / 1e9, b, total / b / 1e9, tu);
      fflush(stdout);
    }
  return 0;
}
