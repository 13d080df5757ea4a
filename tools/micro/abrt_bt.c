// LD_PRELOAD helper for chasing an intermittent abort(): prints the C backtrace of the aborting thread to stderr.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
static void on_abrt(int sig) {
    void* bt[64];
    int n = backtrace(bt, 64);
    const char msg[] = "\n=== SIGABRT backtrace ===\n";
    write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void init(void) { signal(SIGABRT, on_abrt); }
