/* What does writing a fresh multi-GB file in /dev/shm cost, and where?  N threads pwrite 32 MB chunks of a 9 GB file:
 * (a) into a file that does not exist yet (pages allocated by the writes), (b) again into the same file (pages exist),
 * (c) fallocate of a fresh file, then the writes.  gcc -O2 -pthread tools/shm_write_probe.c -o tools/_bin/shm_write_probe */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static const size_t CHUNK = 32u << 20;
static long long total;
static int fd, nthreads;
static long long next_chunk;
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
    if (pwrite(fd, buf, n, off) != (ssize_t)n) { perror("pwrite"); exit(1); }
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
  int counts[] = {1, 4, 8, 16, 32};
  for (int i = 0; i < 5; ++i) {
    nthreads = counts[i];
    unlink(path);
    fd = open(path, O_CREAT | O_RDWR, 0644);
    double a = pass(), b = pass();
    close(fd);
    unlink(path);
    fd = open(path, O_CREAT | O_RDWR, 0644);
    double t0 = now();
    int rc = posix_fallocate(fd, 0, total);
    double f = now() - t0, c = pass();
    close(fd);
    unlink(path);
    printf("%2d threads: fresh file %.2f s (%.1f GB/s)  rewrite %.2f s (%.1f GB/s)  fallocate %.2f s (rc %d) + writes %.2f s (%.1f GB/s)\n", nthreads, a,
           total / a / 1e9, b, total / b / 1e9, f, rc, c, total / c / 1e9);
    fflush(stdout);
  }
  return 0;
}
