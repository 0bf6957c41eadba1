/* GPU box: how fast 16 GiB reach a /dev/shm file -- pwrite() from N threads (one file: the inode lock serialises them; N files)
 * against memcpy() into a MAP_SHARED mapping of the file from N threads (page faults run in parallel).
 * gcc -O2 -pthread -o /tmp/shm_write_test tools/closed_ab/shm_write_test.c && /tmp/shm_write_test */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

static const size_t kPiece = 64u << 20, kTotal = 16ull << 30;
static char *src;
static int fds[64], nt, mode;
static char *map;

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + t.tv_nsec * 1e-9;
}
static void *work(void *arg) {
  const long t = (long)arg;
  const size_t n = kTotal / kPiece;
  for (size_t k = (size_t)t; k < n; k += (size_t)nt) {
    if (mode == 0) {
      if (pwrite(fds[0], src, kPiece, (off_t)(k * kPiece)) != (ssize_t)kPiece) perror("pwrite");
    } else if (mode == 1) {
      if (pwrite(fds[t], src, kPiece, (off_t)(k / (size_t)nt * kPiece)) != (ssize_t)kPiece) perror("pwrite");
    } else {
      memcpy(map + k * kPiece, src, kPiece);
    }
  }
  return NULL;
}
int main(void) {
  src = malloc(kPiece);
  for (size_t i = 0; i < kPiece; i++) src[i] = (char)(i * 2654435761u >> 13);
  const char *names[3] = {"pwrite, one file", "pwrite, one file per thread", "memcpy into mmap(MAP_SHARED), one file"};
  const int counts[5] = {1, 2, 4, 8, 16};
  for (mode = 0; mode < 3; mode++)
    for (int c = 0; c < 5; c++) {
      nt = counts[c];
      char name[64][128];
      const int nf = mode == 1 ? nt : 1;
      for (int f = 0; f < nf; f++) {
        snprintf(name[f], sizeof name[f], "/dev/shm/pbsim_wtest_%d", f);
        fds[f] = open(name[f], O_CREAT | O_RDWR | O_TRUNC, 0644);
      }
      if (mode == 2) {
        if (ftruncate(fds[0], (off_t)kTotal) != 0) perror("ftruncate");
        map = mmap(NULL, kTotal, PROT_READ | PROT_WRITE, MAP_SHARED, fds[0], 0);
        if (map == MAP_FAILED) { perror("mmap"); return 1; }
      }
      pthread_t th[64];
      const double t0 = now();
      for (long t = 0; t < nt; t++) pthread_create(&th[t], NULL, work, (void *)t);
      for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
      const double dt = now() - t0;
      if (mode == 2) munmap(map, kTotal);
      for (int f = 0; f < nf; f++) { close(fds[f]); unlink(name[f]); }
      printf("%-42s %2d threads  %5.1f GB/s\n", names[mode], nt, kTotal / dt / 1e9);
      fflush(stdout);
    }
  return 0;
}
