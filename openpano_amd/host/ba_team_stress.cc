// ba_team_stress.cc -- stress of pano::BaTeam (pano_camera.hh), the spinning team of the bundle adjuster.
// Sections of alternating sizes with (nearly) empty items: the window in which a worker that holds a spent ticket of one
// section could meet the item count of the next is as wide as it gets.  Every item must run exactly once, and a section must
// end; a section that does not end within 20 s is reported (exit 3), a miscount exits 2.
// g++ -std=c++17 -O2 -fopenmp -I openpano_amd/host ba_team_stress.cc -o ba_team_stress; ./ba_team_stress [threads] [sections]
#include <cstdio>
#include <cstdlib>
#include <csignal>
#include <unistd.h>
#include <vector>
#include "pano_camera.hh"

static volatile long g_section = -1;
static void on_alarm(int) { fprintf(stderr, "BaTeam: section %ld did not end\n", g_section); _exit(3); }

int main(int argc, char** argv) {
	const int threads = argc > 1 ? atoi(argv[1]) : 8;
	const long sections = argc > 2 ? atol(argv[2]) : 200000;
	signal(SIGALRM, on_alarm);
	std::vector<std::atomic<int>> hits(4096);
	pano::BaTeam t(threads);
	long bad = 0;
#pragma omp parallel num_threads(threads)
	{
		if (omp_get_thread_num() == 0) {
			t.set_threads(omp_get_num_threads());
			unsigned rng = 12345;
			for (long s = 0; s < sections; ++s) {
				g_section = s;
				if ((s & 1023) == 0) alarm(20);
				rng = rng * 1664525u + 1013904223u;
				// small after large after small: the count of the next section is often above the spent ticket's index
				const int n = (s & 1) ? 2 + (int)((rng >> 8) % 6) : 40 + (int)((rng >> 8) % 2000);
				for (int i = 0; i < n; ++i) hits[i].store(0, std::memory_order_relaxed);
				t.run(n, [&](int i) { hits[i].fetch_add(1, std::memory_order_relaxed); });
				for (int i = 0; i < n; ++i) if (hits[i].load(std::memory_order_relaxed) != 1) ++bad;
			}
			t.finish();
		} else t.worker_loop();
	}
	alarm(0);
	if (bad) { fprintf(stderr, "BaTeam: %ld items ran zero or several times\n", bad); return 2; }
	printf("BaTeam: %ld sections on %d threads, every item once\n", sections, threads);
	return 0;
}
