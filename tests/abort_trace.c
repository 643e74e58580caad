/* Test infrastructure (never linked into the product): names the native thread that calls abort().
 * The rare silent SIGABRT of a -m gpu run leaves only the Python main thread's stack in
 * faulthandler's dump -- wherever that thread happened to be --, while abort() raises the signal on the CALLING
 * thread: this handler prints that thread's name and native backtrace (module + offset per frame) first, then
 * hands over to the handler that was installed before (Python's faulthandler), which dumps and re-raises.
 * Loaded by tests/conftest.py when $SDPA_ABORT_TRACE is set (tools/gpu_flaky_hunt.sh sets it). */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <unistd.h>

static struct sigaction previous;
static int out_fd = 2;      /* a dup of the real stderr taken at install time: pytest captures fd 2 while a test runs */

static void on_abort(int sig, siginfo_t *info, void *ctx) {
    char name[32] = "?", line[160];
    void *frames[96];
    prctl(PR_GET_NAME, name);
    int n = snprintf(line, sizeof line, "\n*** SIGABRT raised on thread '%s' (tid %ld); native backtrace:\n", name,
                     (long)syscall(SYS_gettid));
    if (write(out_fd, line, (size_t)n) < 0) {}
    backtrace_symbols_fd(frames, backtrace(frames, 96), out_fd);
    if (write(out_fd, "*** end of native backtrace\n", 28) < 0) {}
    sigaction(SIGABRT, &previous, NULL);
    if (previous.sa_flags & SA_SIGINFO) {
        if (previous.sa_sigaction) previous.sa_sigaction(sig, info, ctx);
    } else if (previous.sa_handler != SIG_DFL && previous.sa_handler != SIG_IGN) {
        previous.sa_handler(sig);
    }
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}

int abort_trace_install(int fd) {
    struct sigaction sa;
    void *warm[4];
    if (fd >= 0) out_fd = fd;
    backtrace(warm, 4);               /* loads libgcc's unwinder now, not inside the handler */
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_abort;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    return sigaction(SIGABRT, &sa, &previous);
}
