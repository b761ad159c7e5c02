/* LD_PRELOAD helper: a C backtrace on SIGABRT / SIGSEGV that arrives after Python's faulthandler is gone (process teardown).
 * gcc -shared -fPIC -O1 tools/abort_bt.c -o ab/libabortbt.so ; LD_PRELOAD=$PWD/ab/libabortbt.so python ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig)
{
    void* frames[64];
    const char* msg = sig == SIGABRT ? "\n=== abort_bt: SIGABRT, C backtrace ===\n" : "\n=== abort_bt: SIGSEGV, C backtrace ===\n";
    write(2, msg, strlen(msg));
    alarm(10);                                  /* a corrupted heap can deadlock the unwinder: never hang the box */
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    void* warm[4];
    backtrace(warm, 4);                          /* loads libgcc_s NOW: the first backtrace() call allocates, the handler must not */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler;
    sigaction(SIGABRT, &sa, 0);
}
