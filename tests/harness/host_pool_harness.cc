// Stress harness of openpano_amd/csrc/host_pool.hpp (compiled by tests/test_host_pool_cpu.py with g++, once plain and once with
// -fsanitize=thread): thousands of parallel loops of random length and random item cost, from two caller threads, on a
// pool with more workers than this machine has cores.  Every item of every loop must run exactly once, inside its loop
// (never after run() has returned), with the body of ITS loop.
#include "host_pool.hpp"
#include <cstdio>
#include <random>
#include <vector>
#include <time.h>

int main(int argc, char** argv) {
	const int loops = argc > 1 ? atoi(argv[1]) : 20000;
	ophost::HostPool& pool = ophost::host_pool();
	std::atomic<long> errors{0}, items{0};
	auto caller = [&](unsigned seed) {
		std::mt19937 rng(seed);
		std::vector<std::atomic<int>> hits(4096);
		for (int l = 0; l < loops; ++l) {
			const int n = 2 + (int)(rng() % (l % 16 == 0 ? 4000 : 70));
			const int grain = 1 + (int)(rng() % 5);
			const int cost = (int)(rng() % 200);
			for (int i = 0; i < n; ++i) hits[i].store(0, std::memory_order_relaxed);
			std::atomic<bool> open{true};
			const int tag = l;
			const std::function<void(int)> body = [&, tag](int i) {
				if (!open.load(std::memory_order_acquire) || i < 0 || i >= n || tag != l) errors.fetch_add(1);
				volatile double x = 1.0;
				for (int k = 0; k < cost * (1 + (i & 3)); ++k) x = x * 1.0000001 + 1e-9;
				hits[i].fetch_add(1, std::memory_order_relaxed);
			};
			pool.run(n, body, grain);
			open.store(false, std::memory_order_release);
			for (int i = 0; i < n; ++i) if (hits[i].load(std::memory_order_relaxed) != 1) errors.fetch_add(1);
			items.fetch_add(n);
		}
	};
	timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
	std::thread other(caller, 777u);
	caller(4242u);
	other.join();
	clock_gettime(CLOCK_MONOTONIC, &t1);
	std::printf("workers %d loops %d items %ld errors %ld seconds %.2f\n", pool.nworkers, 2 * loops, items.load(), errors.load(),
			(t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
	return errors.load() == 0 ? 0 : 1;
}
