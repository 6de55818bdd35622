/* Diagnostic LD_PRELOAD shim (not part of the product): on SIGSEGV print what a post-mortem needs - the faulting address, whether the
 * access was a read or a write, the instruction pointer, the /proc/self/maps lines around both, and every frame of the
 * backtrace as module + offset (symbolised offline with addr2line / nm) - then hand over to whatever handler the process
 * installed afterwards (rocprofv3's tool installs glog's).  Used once in round 3 to find the abort recorded in
 * gpurun_out/bclock30_prof.log.
 *   gcc -O1 -g -shared -fPIC -o tools/segv_maps.so tools/segv_maps.c -ldl
 *   LD_PRELOAD=$PWD/tools/segv_maps.so rocprofv3 --kernel-trace ... -- python3 tools/gpu_batch_clock.py 30 40            */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <unistd.h>

static struct sigaction g_next;
static int g_have_next = 0, g_installed = 0;
static int (*g_real_sigaction)(int, const struct sigaction*, struct sigaction*) = 0;

static void print_maps_near(unsigned long a, const char* what) {
  FILE* f = fopen("/proc/self/maps", "r");
  if (!f) return;
  char prev[512] = "", line[512];
  int after = 0;
  dprintf(2, "[segv_maps] mappings around the %s %#lx:\n", what, a);
  while (fgets(line, sizeof(line), f)) {
    unsigned long lo = 0, hi = 0;
    if (sscanf(line, "%lx-%lx", &lo, &hi) != 2) continue;
    if (after > 0) { dprintf(2, "    next : %s", line); if (--after == 0) break; continue; }
    if (a >= lo && a < hi) { dprintf(2, "    prev : %s    IN   : %s", prev, line); after = 2; }
    else if (lo > a) { dprintf(2, "    prev : %s    (address %#lx is in NO mapping; the gap ends at)\n    next : %s", prev, a, line); after = 1; }
    strncpy(prev, line, sizeof(prev) - 1);
  }
  fclose(f);
}

static void handler(int sig, siginfo_t* si, void* ucv) {
  ucontext_t* uc = (ucontext_t*)ucv;
  const unsigned long err = (unsigned long)uc->uc_mcontext.gregs[REG_ERR], rip = (unsigned long)uc->uc_mcontext.gregs[REG_RIP];
  dprintf(2, "\n[segv_maps] signal %d, address %p, %s access, si_code %d, rip %#lx, rsp %#lx, rsi %#lx, rdi %#lx, rdx %#lx, rcx %#lx, tid %ld\n",
          sig, si->si_addr, (err & 2) ? "WRITE" : "READ", si->si_code, rip, (unsigned long)uc->uc_mcontext.gregs[REG_RSP],
          (unsigned long)uc->uc_mcontext.gregs[REG_RSI], (unsigned long)uc->uc_mcontext.gregs[REG_RDI],
          (unsigned long)uc->uc_mcontext.gregs[REG_RDX], (unsigned long)uc->uc_mcontext.gregs[REG_RCX], (long)gettid());
  print_maps_near((unsigned long)si->si_addr, "faulting address");
  print_maps_near((unsigned long)uc->uc_mcontext.gregs[REG_RSP], "stack pointer");
  void* bt[64];
  const int n = backtrace(bt, 64);
  for (int i = 0; i < n; ++i) {
    Dl_info di;
    if (dladdr(bt[i], &di) && di.dli_fname)
      dprintf(2, "[segv_maps] #%02d %p  %s + %#lx  (%s)\n", i, bt[i], di.dli_fname, (unsigned long)((char*)bt[i] - (char*)di.dli_fbase),
              di.dli_sname ? di.dli_sname : "?");
    else dprintf(2, "[segv_maps] #%02d %p  ?\n", i, bt[i]);
  }
  if (g_have_next && (g_next.sa_flags & SA_SIGINFO) && g_next.sa_sigaction) { g_next.sa_sigaction(sig, si, ucv); return; }
  if (g_have_next && g_next.sa_handler && g_next.sa_handler != SIG_DFL && g_next.sa_handler != SIG_IGN) { g_next.sa_handler(sig); return; }
  signal(sig, SIG_DFL);
  raise(sig);
}

int sigaction(int signum, const struct sigaction* act, struct sigaction* old) {
  if (!g_real_sigaction) g_real_sigaction = (int (*)(int, const struct sigaction*, struct sigaction*))dlsym(RTLD_NEXT, "sigaction");
  if (signum == SIGSEGV && g_installed && act) {          /* somebody installs a handler after us: remember it, stay first */
    if (old) { if (g_have_next) *old = g_next; else memset(old, 0, sizeof(*old)); }
    g_next = *act;
    g_have_next = 1;
    return 0;
  }
  return g_real_sigaction(signum, act, old);
}

__attribute__((constructor)) static void segv_maps_init(void) {
  if (!g_real_sigaction) g_real_sigaction = (int (*)(int, const struct sigaction*, struct sigaction*))dlsym(RTLD_NEXT, "sigaction");
  void* warm[4];
  (void)backtrace(warm, 4);                                /* loads libgcc now, not inside the handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO;
  sigemptyset(&sa.sa_mask);
  if (g_real_sigaction(SIGSEGV, &sa, &g_next) == 0) {
    g_installed = 1;
    g_have_next = g_next.sa_handler != SIG_DFL && g_next.sa_handler != SIG_IGN;
  }
}
