/* abort_bt.c — TEST INFRASTRUCTURE (never linked into the product).
 *
 * A SIGABRT handler that writes the NATIVE backtrace of the aborting thread to a file of its own and to the process's
 * original stderr, then lets the abort proceed.  Why: pytest captures fd 2 per test, so whatever the HIP runtime or
 * libstdc++ printed before calling abort() is lost with the capture file when the process dies — the "silent abort()"
 * of rounds 4 and 5.  tests/conftest.py loads this when RR_ABORT_BT=1.
 *
 *   gcc -O1 -g -shared -fPIC -o abort_bt.so abort_bt.c
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static int g_err_fd = -1;
static char g_path[512];
static struct sigaction g_prev;

static void put(int fd, const char* s) { if (fd >= 0) (void)!write(fd, s, strlen(s)); }

static void on_abort(int sig) {
    void* frames[96];
    int n = backtrace(frames, 96);
    int fd = open(g_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    int fds[2] = {fd, g_err_fd};
    for (int k = 0; k < 2; k++) {
        put(fds[k], "\n=== SIGABRT: native backtrace of the aborting thread ===\n");
        if (fds[k] >= 0) backtrace_symbols_fd(frames, n, fds[k]);
        put(fds[k], "=== end of backtrace ===\n");
    }
    if (fd >= 0) {   /* the mappings, to turn module+offset into symbols afterwards */
        int m = open("/proc/self/maps", O_RDONLY);
        if (m >= 0) {
            char buf[4096];
            ssize_t r;
            put(fd, "=== /proc/self/maps (r-x only would do; all kept) ===\n");
            while ((r = read(m, buf, sizeof buf)) > 0) (void)!write(fd, buf, (size_t)r);
            close(m);
        }
        close(fd);
    }
    sigaction(sig, &g_prev, NULL);   /* whoever was there before (Python's faulthandler: the Python stacks), then the default */
    raise(sig);
}

/* path: file the backtrace is appended to.  Call BEFORE anything starts capturing fd 2 if the stderr copy is wanted. */
int abort_bt_install(const char* path) {
    void* warm[4];
    (void)backtrace(warm, 4);                      /* loads libgcc_s now: no dlopen inside the handler */
    strncpy(g_path, path, sizeof g_path - 1);
    if (g_err_fd < 0) g_err_fd = dup(2);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_abort;
    sa.sa_flags = SA_NODEFER;
    return sigaction(SIGABRT, &sa, &g_prev);
}
