// LD_PRELOAD helper: print a native backtrace when the process receives SIGABRT (debugging silent runtime aborts on the GPU box).
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <string.h>
static void on_abort(int sig) {
	void *frames[96];
	const char msg[] = "\n==== SIGABRT: native backtrace ====\n";
	write(2, msg, sizeof(msg) - 1);
	int n = backtrace(frames, 96);
	backtrace_symbols_fd(frames, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}
__attribute__((constructor)) static void install(void) { signal(SIGABRT, on_abort); }
