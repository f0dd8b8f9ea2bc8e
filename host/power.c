/* power.c -- see power.h.  Own code; the reference's sampler is a shell one-liner around nvidia-smi. */
#define _GNU_SOURCE
#include "power.h"

#include <ctype.h>
#include <glob.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct power_sampler {
  char node[512];
  FILE *out;
  int period_ms;
  volatile int stop;
  pthread_t thread;
  double sum_watts, energy;
  int n;
};

static double mono_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int read_microwatts(const char *node, double *uw) {
  FILE *f = fopen(node, "r");
  if (!f) return 1;
  long long v = 0;
  const int ok = fscanf(f, "%lld", &v) == 1;
  fclose(f);
  if (!ok) return 1;
  *uw = (double)v;
  return 0;
}

static void *sampler_main(void *arg) {
  power_sampler *s = (power_sampler *)arg;
  double last = mono_s();
  while (!s->stop) {
    double uw;
    if (read_microwatts(s->node, &uw) == 0) {
      const double w = uw * 1e-6, now = mono_s();
      fprintf(s->out, "%.3f\n", w);
      s->sum_watts += w;
      s->energy += w * (now - last);
      last = now;
      s->n++;
    }
    struct timespec d = {s->period_ms / 1000, (long)(s->period_ms % 1000) * 1000000L};
    nanosleep(&d, NULL);
  }
  return NULL;
}

static int find_node(const char *pci_bus_id, char *node, size_t cap) {
  char lower[64];
  size_t k = 0;
  for (; pci_bus_id[k] && k + 1 < sizeof lower; k++) lower[k] = (char)tolower((unsigned char)pci_bus_id[k]);
  lower[k] = 0;
  static const char *names[] = {"power1_average", "power1_input"};
  for (int i = 0; i < 2; i++) {
    char pat[256];
    snprintf(pat, sizeof pat, "/sys/bus/pci/devices/%s/hwmon/hwmon*/%s", lower, names[i]);
    glob_t g;
    if (glob(pat, 0, NULL, &g) == 0 && g.gl_pathc > 0) {
      double uw;
      if (read_microwatts(g.gl_pathv[0], &uw) == 0) {
        snprintf(node, cap, "%s", g.gl_pathv[0]);
        globfree(&g);
        return 0;
      }
    }
    globfree(&g);
  }
  return 1;
}

int power_sampler_start(const char *pci_bus_id, const char *out_path, int period_ms, power_sampler **out) {
  *out = NULL;
  power_sampler *s = (power_sampler *)calloc(1, sizeof *s);
  if (!s) return 1;
  if (!pci_bus_id || find_node(pci_bus_id, s->node, sizeof s->node) != 0) {
    fprintf(stderr, "power sampling: no readable hwmon power node for device %s\n", pci_bus_id ? pci_bus_id : "(null)");
    free(s);
    return 1;
  }
  s->out = fopen(out_path, "a");   /* the reference appends (>>) as well */
  if (!s->out) {
    fprintf(stderr, "power sampling: cannot open %s\n", out_path);
    free(s);
    return 1;
  }
  s->period_ms = period_ms > 0 ? period_ms : 100;
  if (pthread_create(&s->thread, NULL, sampler_main, s) != 0) {
    fclose(s->out);
    free(s);
    return 1;
  }
  *out = s;
  return 0;
}

int power_sampler_stop(power_sampler *s, double *mean_watts, double *energy_joules, int *nsamples) {
  if (!s) return 1;
  s->stop = 1;
  pthread_join(s->thread, NULL);
  fclose(s->out);
  if (mean_watts) *mean_watts = s->n ? s->sum_watts / s->n : 0.0;
  if (energy_joules) *energy_joules = s->energy;
  if (nsamples) *nsamples = s->n;
  free(s);
  return 0;
}
