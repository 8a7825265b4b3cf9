// development aid: print a native backtrace on SIGSEGV (LD_PRELOAD or ctypes.CDLL before the failing call)
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig, siginfo_t* si, void* uc) {
  void* bt[64];
  const char msg[] = "\n[segv_bt] signal caught, backtrace:\n";
  write(2, msg, sizeof(msg) - 1);
  int n = backtrace(bt, 64);
  backtrace_symbols_fd(bt, n, 2);
  _exit(139);
}
__attribute__((constructor)) static void init(void) {
  static char stack[1 << 16];
  stack_t ss; ss.ss_sp = stack; ss.ss_size = sizeof(stack); ss.ss_flags = 0;
  sigaltstack(&ss, 0);
  struct sigaction sa; memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = handler; sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, 0); sigaction(SIGBUS, &sa, 0);
}
