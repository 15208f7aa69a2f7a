// LD_PRELOAD sampling profiler: SIGPROF every 1 ms of process CPU time, records the interrupted PC; at exit writes
// "<pc> <count>" lines plus /proc/self/maps to $PCSAMPLE_OUT.  Resolve with tools/scratch/pcsample_report.py.
#define _GNU_SOURCE
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>
#include <unistd.h>
#include <execinfo.h>
#define CAP (1 << 22)
static unsigned long* buf; static volatile long n;
static int depth;
static void on_prof(int sig, siginfo_t* si, void* uc) {
  long i = __sync_fetch_and_add(&n, 1); if (i >= CAP / 8) return;
  void* fr[10]; int k = depth ? backtrace(fr, 10) : 0;
  unsigned long* b = buf + i * 8; b[0] = ((ucontext_t*)uc)->uc_mcontext.gregs[REG_RIP];
  for (int j = 1; j < 8; j++) b[j] = (j + 2 < k) ? (unsigned long)fr[j + 2] : 0;   // fr[0]=handler, fr[1]=restorer, fr[2]=pc
}
static void dump(void) {
  struct itimerval z = {{0, 0}, {0, 0}}; setitimer(ITIMER_PROF, &z, 0);
  const char* o = getenv("PCSAMPLE_OUT"); if (!o) o = "/tmp/pcsample.out";
  char path[512]; snprintf(path, sizeof path, "%s.%d", o, (int)getpid()); if (n < 50) return; FILE* f = fopen(path, "w"); if (!f) return;
  long m = n < CAP / 8 ? n : CAP / 8;
  for (long i = 0; i < m; i++) { for (int j = 0; j < 8; j++) fprintf(f, "%lx ", buf[i * 8 + j]); fprintf(f, "\n"); }
  fprintf(f, "MAPS\n");
  FILE* mp = fopen("/proc/self/maps", "r"); char line[1024];
  while (mp && fgets(line, sizeof line, mp)) if (strstr(line, " r-xp ") || strstr(line, "r-xp")) fputs(line, f);
  fclose(f);
}
__attribute__((constructor)) static void init(void) {
  buf = calloc(CAP, sizeof(unsigned long)); depth = getenv("PCSAMPLE_STACK") != 0; { void* t[4]; backtrace(t, 4); }
  struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART; sigaction(SIGPROF, &sa, 0);
  struct itimerval it = {{0, 1000}, {0, 1000}}; setitimer(ITIMER_PROF, &it, 0);
  atexit(dump);
}
