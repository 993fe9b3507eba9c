/* LD_PRELOAD helper (diagnostics only): print a native backtrace when the process receives SIGABRT / SIGSEGV, so a glibc heap-check
   abort at process exit names the destructor it happened in.  gcc -shared -fPIC -o abrt_bt.so abrt_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_sig(int sig)
{
    void* bt[96];
    const char* m = sig == SIGABRT ? "\n[abrt_bt] SIGABRT backtrace:\n" : "\n[abrt_bt] SIGSEGV backtrace:\n";
    (void)!write(2, m, strlen(m));
    int n = backtrace(bt, 96);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void init(void)
{
    void* bt[4];
    backtrace(bt, 4);              /* loads libgcc now, not inside the handler */
    signal(SIGABRT, on_sig);
    signal(SIGSEGV, on_sig);
}
