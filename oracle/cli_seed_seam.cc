// oracle/cli_seed_seam.cc -- TEST INFRASTRUCTURE ONLY.
//
// Determinism seam of the CLI builds (_ref/image-stitching, _ref/image-stitching-hip*): the reference seeds every
// TransformEstimation::get_transform from std::random_device (stitch/transform_estimate.cc:64-65), and so does the
// HIP adapter (openpano_amd/host/pano_hip.hh).  libstdc++ implements random_device::operator() through the
// out-of-line member _M_getval(); this definition, linked into the executables, returns OPENPANO_TEST_SEED instead
// of entropy, so that a CPU run and a hooked run of the reference's own main.cc draw the same samples.
#include <cstdlib>
#include <random>
namespace std {
unsigned int random_device::_M_getval() {
	static const unsigned seed = [] { const char* e = getenv("OPENPANO_TEST_SEED"); return e ? (unsigned)strtoul(e, nullptr, 10) : 20260922u; }();
	return seed;
}
}

// A crash of a CLI build under test should say where: SIGSEGV / SIGABRT print a backtrace (symbol names come from
// -rdynamic) before the default action.
#include <csignal>
#include <execinfo.h>
#include <unistd.h>
namespace {
void crash_trace(int sig) {
	void* frames[64];
	const int n = backtrace(frames, 64);
	const char msg[] = "\n[cli_seed_seam] fatal signal, backtrace:\n";
	if (write(2, msg, sizeof(msg) - 1) < 0) {}
	backtrace_symbols_fd(frames, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}
struct InstallTrace { InstallTrace() { signal(SIGSEGV, crash_trace); signal(SIGABRT, crash_trace); signal(SIGBUS, crash_trace); } } g_install_trace;
}
