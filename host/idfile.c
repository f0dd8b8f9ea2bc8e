/*
 * idfile.c -- hand-over of the 256-byte communicator id from rank 0 to the other ranks through a file on a directory
 * all ranks see (driver flag -id_file).  Stands in for the one MPI_Bcast an MPI build would use (solvers.c:232-238
 * broadcasts its sizes the same way); the driver itself needs no MPI.
 *
 * File layout: 256 bytes of id, then 64 bytes of launch nonce (text, zero padded).  A file left behind by an earlier
 * run that died (rank 0 removes it on every exit path it controls, a kill leaves it) must not be taken for this run's:
 *   - rank 0 removes any file of that name before it draws the id, and writes under another name + rename, so a reader
 *     never sees half a file;
 *   - every rank derives the same nonce from the launcher's environment (MK_LAUNCH_ID, SLURM_JOB_ID[.SLURM_STEP_ID],
 *     PMIX_NAMESPACE, PMI_JOBID, MASTER_ADDR:MASTER_PORT) and a reader takes only a file that carries its own nonce;
 *   - without any of those variables a reader takes only a file younger than max_age_s seconds (default 120) before
 *     its own start: older ones are leftovers.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "sparse_matrix.h"

#define MK_ID_BYTES 256
#define MK_NONCE_BYTES 64

void mk_idfile_nonce(char *nonce64) {
  memset(nonce64, 0, MK_NONCE_BYTES);
  const char *v;
  if ((v = getenv("MK_LAUNCH_ID")) && *v) { snprintf(nonce64, MK_NONCE_BYTES, "L:%s", v); return; }
  if ((v = getenv("SLURM_JOB_ID")) && *v) {
    const char *s = getenv("SLURM_STEP_ID");
    snprintf(nonce64, MK_NONCE_BYTES, "S:%s.%s", v, s ? s : "");
    return;
  }
  if ((v = getenv("PMIX_NAMESPACE")) && *v) { snprintf(nonce64, MK_NONCE_BYTES, "X:%s", v); return; }
  if ((v = getenv("PMI_JOBID")) && *v) { snprintf(nonce64, MK_NONCE_BYTES, "P:%s", v); return; }
  if ((v = getenv("MASTER_PORT")) && *v) {
    const char *a = getenv("MASTER_ADDR");
    snprintf(nonce64, MK_NONCE_BYTES, "M:%s:%s", a ? a : "", v);
    return;
  }
}

/* rank 0, step 1: BEFORE anything slow (and before the id exists) take a leftover out of the readers' way */
void mk_idfile_clear(const char *path) {
  char tmp[4096];
  remove(path);
  snprintf(tmp, sizeof tmp, "%s.tmp", path);
  remove(tmp);
}

/* rank 0, step 2 */
int mk_idfile_publish(const char *path, const unsigned char *id256) {
  char tmp[4096], nonce[MK_NONCE_BYTES];
  mk_idfile_nonce(nonce);
  snprintf(tmp, sizeof tmp, "%s.tmp", path);
  FILE *f = fopen(tmp, "wb");
  if (!f) return 1;
  const int ok = fwrite(id256, 1, MK_ID_BYTES, f) == MK_ID_BYTES && fwrite(nonce, 1, MK_NONCE_BYTES, f) == MK_NONCE_BYTES;
  if (fclose(f) != 0 || !ok || rename(tmp, path) != 0) { remove(tmp); return 1; }
  return 0;
}

/* ranks > 0: wait up to timeout_s for THIS launch's id; 0 = got it, 1 = timed out, 2 = only leftovers were seen */
int mk_idfile_wait(const char *path, unsigned char *id256, double timeout_s, double max_age_s) {
  char nonce[MK_NONCE_BYTES], got_nonce[MK_NONCE_BYTES];
  unsigned char buf[MK_ID_BYTES];
  mk_idfile_nonce(nonce);
  const time_t started = time(NULL);
  int stale_seen = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (;;) {
    FILE *f = fopen(path, "rb");
    if (f) {
      struct stat sb;
      const int have_stat = fstat(fileno(f), &sb) == 0;
      const size_t a = fread(buf, 1, MK_ID_BYTES, f), b = fread(got_nonce, 1, MK_NONCE_BYTES, f);
      fclose(f);
      if (a == MK_ID_BYTES && b == MK_NONCE_BYTES) {
        int mine;
        if (nonce[0]) mine = memcmp(nonce, got_nonce, MK_NONCE_BYTES) == 0;
        else mine = got_nonce[0] == 0 && have_stat && difftime(started, sb.st_mtime) <= max_age_s;
        if (mine) { memcpy(id256, buf, MK_ID_BYTES); return 0; }
        stale_seen = 1;
      }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s) return stale_seen ? 2 : 1;
    usleep(50000);
  }
}
