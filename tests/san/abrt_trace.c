/* abrt_trace.c -- test infrastructure: prints the C call stack of whoever raises SIGABRT (glibc's heap checks, std::terminate,
 * a runtime's abort()) to stderr, then hands over to the handler that was installed before (Python's faulthandler, which
 * shows the Python frames only) or to the default action.  tests/conftest.py builds and loads it (abrt_trace_install) on
 * every test run, so that an abort in a GPU test run leaves evidence; also usable as LD_PRELOAD.  Never loaded by the product. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <unistd.h>

/* When the shim is LD_PRELOADed it also sees every pthread_create of the process and remembers each thread's start routine:
 * the abort hunted here is raised on a thread whose start routine jumps straight into abort(), so the call stack names nobody. */
#include <pthread.h>
#include <stdlib.h>
typedef struct { void *(*fn)(void *); void *arg; } start_t;
static struct { long tid; void *fn; } started[4096];
static volatile int n_started;
static void *start_shim(void *p)
{
    start_t s = *(start_t *)p;
    free(p);
    const int k = __sync_fetch_and_add(&n_started, 1);
    if (k < 4096) { started[k].tid = (long)syscall(SYS_gettid); started[k].fn = (void *)s.fn; }
    return s.fn(s.arg);
}
int pthread_create(pthread_t *th, const pthread_attr_t *attr, void *(*fn)(void *), void *arg)
{
    static int (*real)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *);
    if (!real) real = (int (*)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *))dlsym(RTLD_NEXT, "pthread_create");
    start_t *s = (start_t *)malloc(sizeof(start_t));
    if (!s) return real(th, attr, fn, arg);
    s->fn = fn; s->arg = arg;
    return real(th, attr, start_shim, s);
}

static struct sigaction prev_sa;
static int out_fd = 2;
static volatile sig_atomic_t fired;

static void on_abrt(int sig)
{
    void *frames[64];
    static const char head[] = "\n=== abrt_trace: SIGABRT, C call stack of the raising thread ===\n";
    if (fired) { signal(sig, SIG_DFL); raise(sig); return; }      /* second time round (handlers chained to each other): die */
    fired = 1;
    (void)!write(out_fd, head, sizeof(head) - 1);
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, out_fd);
    {   /* who is this thread, and what did it run lately: the words on its stack that point into loaded code (stale return
         * addresses included) -- the unwinder above stops where a frame was left by a jump instead of a call */
        char name[32] = {0}, line[512];
        prctl(PR_GET_NAME, name, 0, 0, 0);
        int len = snprintf(line, sizeof(line), "--- thread %ld \"%s\"; code addresses on its stack, innermost first ---\n", (long)syscall(SYS_gettid), name);
        (void)!write(out_fd, line, (size_t)len);
        {   /* its start routine, if the shim saw the thread being created (LD_PRELOAD) */
            const long me = (long)syscall(SYS_gettid);
            const int ns = n_started < 4096 ? n_started : 4096;
            for (int k = 0; k < ns; k++)
                if (started[k].tid == me) {
                    Dl_info fi; memset(&fi, 0, sizeof(fi));
                    (void)dladdr(started[k].fn, &fi);
                    len = snprintf(line, sizeof(line), "--- its start routine: %s(+0x%lx) %s ---\n", fi.dli_fname ? fi.dli_fname : "?",
                                   (unsigned long)((uintptr_t)started[k].fn - (uintptr_t)fi.dli_fbase), fi.dli_sname ? fi.dli_sname : "");
                    (void)!write(out_fd, line, (size_t)len);
                }
            len = snprintf(line, sizeof(line), "--- threads created while the shim was watching: %d ---\n", ns);
            (void)!write(out_fd, line, (size_t)len);
        }
        uintptr_t *sp = (uintptr_t *)__builtin_frame_address(0);
        int shown = 0;
        for (int i = 0; i < 6000 && shown < 160; i++) {
            /* stay inside the mapped stack: a page that is not there ends the scan (mincore) */
            if (((uintptr_t)(sp + i) & 4095) == 0 || i == 0) {
                unsigned char v;
                if (syscall(SYS_mincore, (void *)((uintptr_t)(sp + i) & ~(uintptr_t)4095), 4096, &v) != 0) break;
            }
            Dl_info info;
            const uintptr_t a = sp[i];
            if (a < 0x10000 || !dladdr((void *)a, &info) || !info.dli_fname) continue;
            len = snprintf(line, sizeof(line), "  [sp+%d] %s(+0x%lx) %s\n", i * 8, info.dli_fname, (unsigned long)(a - (uintptr_t)info.dli_fbase), info.dli_sname ? info.dli_sname : "");
            (void)!write(out_fd, line, (size_t)len);
            shown++;
        }
    }
    sigaction(sig, &prev_sa, 0);      /* faulthandler's (or the default): it gets the re-raised signal */
    raise(sig);
}

void abrt_trace_install(int fd)       /* fd: where to write (a duplicate of the real stderr: the test runner redirects fd 2) */
{
    if (fd >= 0) out_fd = fd;
    struct sigaction cur;
    if (sigaction(SIGABRT, 0, &cur) == 0 && cur.sa_handler == on_abrt) return;      /* already on top */
    void *warm[2];
    backtrace(warm, 2);               /* loads libgcc now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_abrt;
    sa.sa_flags = SA_NODEFER;
    sigaction(SIGABRT, &sa, &prev_sa);
}

__attribute__((constructor)) static void on_load(void)
{
    /* as LD_PRELOAD the shim is in place before anything else; loaded from conftest.py the explicit call re-installs it on
     * top of whatever the test runner put there in the meantime */
    abrt_trace_install(-1);
}
