// types_selftest.cc -- standalone value types (pano_types.hh): MatchInfo text round trip.
//   types_selftest <confidence> <9 homo> <n> <4n coords>   -> prints MatchInfo::serialize(),
//   then dumps/loads it through dump_matchinfo / load_matchinfo and prints the reloaded form.
// No GPU and no libopenpano_hip needed; tests/test_host_types_cpu.py compares the first line with
// the reference's own MatchInfo::serialize (oracle/_ref).
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include "pano_types.hh"
using namespace pano;
int main(int argc, char** argv) {
	if (argc < 12) return 2;
	MatchInfo m;
	m.confidence = (float)atof(argv[1]);
	for (int i = 0; i < 9; ++i) m.homo[i] = atof(argv[2 + i]);
	const int n = atoi(argv[11]);
	if (argc < 12 + 4 * n) return 2;
	for (int i = 0; i < n; ++i)
		m.match.emplace_back(Vec2D(atof(argv[12 + 4 * i]), atof(argv[13 + 4 * i])), Vec2D(atof(argv[14 + 4 * i]), atof(argv[15 + 4 * i])));
	m.serialize(std::cout); std::cout << std::endl;
	std::vector<std::vector<MatchInfo>> pm(3, std::vector<MatchInfo>(3));
	pm[0][2] = m; pm[2][0] = m; pm[2][0].reverse();
	const char* f = argc > 12 + 4 * n ? argv[12 + 4 * n] : "/tmp/matchinfo_selftest.txt";
	dump_matchinfo(f, pm);
	auto back = load_matchinfo(f, 3);
	back[0][2].serialize(std::cout); std::cout << std::endl;
	std::cout << (back[1][1].confidence == 0 && back[2][0].match.size() == m.match.size()) << std::endl;
	return 0;
}
