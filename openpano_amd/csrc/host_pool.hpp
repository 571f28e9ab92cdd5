// host_pool.hpp -- the host thread pool behind host_parallel_for (capi.hip).  Plain C++ / Linux, no HIP: the stress harness
// tests/harness/host_pool_harness.cc compiles it with g++ (and with -fsanitize=thread).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <functional>
// A parallel-for whose start costs what waking its threads in PARALLEL costs.  The first form parked the workers on a
// condition variable: notify_all() makes every woken thread take the mutex in turn, ~3.5 us each -- 0.22 ms before the
// last of 63 workers had started on a loop that is 0.06 ms of work per thread (the RANSAC acceptance epilogue; measured
// with OPENPANO_HOST_THREADS = 1 .. 64, profiles/r06_ransac_threads.txt).  Now the workers sleep on a futex word; the
// caller wakes two of them, every woken worker wakes two more before it starts on the items (a binary tree: six levels for
// 63 workers), and only as many are woken as the loop has use for.  Items are claimed by compare-exchange on one 64-bit
// word (loop number, next index): a worker that comes late -- still on its way back from the previous loop -- can never
// claim an item of a loop it has not read the description of, and the caller waits for the ITEMS (a count), not for threads.
#include <atomic>
#include <mutex>
#include <thread>
#include <climits>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
namespace ophost {
struct HostPool {
	struct Loop { std::atomic<const std::function<void(int)>*> body{nullptr}; std::atomic<int> n{0}; };
	Loop desc[2];                                              // description of loop g in desc[g & 1]: rewritten for g + 2 only after g + 1 has ended
	alignas(64) std::atomic<unsigned long long> ticket{0};     // (loop number << 32) | next unclaimed index
	alignas(64) std::atomic<int> done{0};                      // items of the current loop that have finished
	alignas(64) std::atomic<unsigned> wake_gen{0};             // the futex word: changes once per loop
	std::atomic<int> to_wake{0};                               // sleepers still to be woken for the current loop
	unsigned loop_no = 0;
	int nworkers = 0;
	std::mutex run_mu;      // one parallel loop at a time
	static long futex(std::atomic<unsigned>* w, int op, unsigned val) { return syscall(SYS_futex, (unsigned*)w, op | FUTEX_PRIVATE_FLAG, val, nullptr, nullptr, 0); }
	HostPool() {
		unsigned hw = std::thread::hardware_concurrency();
		int nt = (int)std::min<unsigned>(hw ? hw : 4, 64) - 1;
		if (const char* v = getenv("OPENPANO_HOST_THREADS")) {              // the integrator's cap: threads of a host loop, the caller included
			const int want = atoi(v);
			if (want >= 1) nt = std::min(want, 256) - 1;
		}
		nworkers = nt;
		for (int i = 0; i < nt; ++i) std::thread([this] { worker(); }).detach();     // never joined: the pool lives as long as the process
	}
	// claim and run items of the loop the ticket names; returns when that loop has none left
	void drain() {
		for (;;) {
			unsigned long long t = ticket.load(std::memory_order_acquire);
			const Loop& d = desc[(t >> 32) & 1];                              // read BEFORE the claim: valid if the claim succeeds (see above)
			const int dn = d.n.load(std::memory_order_relaxed);
			const std::function<void(int)>* db = d.body.load(std::memory_order_relaxed);
			const int i = (int)(t & 0xFFFFFFFFu);
			if (i >= dn) return;
			if (!ticket.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel, std::memory_order_relaxed)) continue;
			(*db)(i);
			done.fetch_add(1, std::memory_order_release);
		}
	}
	void wake_two() {
		int k = to_wake.load(std::memory_order_relaxed);
		while (k > 0 && !to_wake.compare_exchange_weak(k, k - std::min(k, 2), std::memory_order_relaxed)) {}
		if (k > 0) futex(&wake_gen, FUTEX_WAKE, (unsigned)std::min(k, 2));
	}
	void worker() {
		unsigned seen = wake_gen.load(std::memory_order_acquire);
		for (;;) {
			while (wake_gen.load(std::memory_order_acquire) == seen) futex(&wake_gen, FUTEX_WAIT, seen);
			seen = wake_gen.load(std::memory_order_acquire);
			wake_two();
			drain();
		}
	}
	void run(int count, const std::function<void(int)>& f, int grain) {
		std::lock_guard<std::mutex> rl(run_mu);
		++loop_no;
		desc[loop_no & 1].body.store(&f, std::memory_order_relaxed); desc[loop_no & 1].n.store(count, std::memory_order_relaxed);
		done.store(0, std::memory_order_relaxed);
		ticket.store((unsigned long long)loop_no << 32, std::memory_order_release);
		to_wake.store(std::min(nworkers, (count + grain - 1) / grain), std::memory_order_relaxed);   // a thread per `grain` items at most: waking one costs microseconds
		wake_gen.fetch_add(1, std::memory_order_release);
		wake_two();
		drain();
		for (unsigned spins = 0; done.load(std::memory_order_acquire) < count; ++spins) {     // the items other threads are still in
#if defined(__x86_64__)
			__builtin_ia32_pause();
#endif
			if ((spins & 0xFFFu) == 0xFFFu) std::this_thread::yield();
		}
		to_wake.store(0, std::memory_order_relaxed);
	}
};
inline HostPool& host_pool() { static HostPool* p = new HostPool; return *p; }     // leaked: no join at exit
}	// namespace ophost
