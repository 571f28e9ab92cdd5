// oracle/ref_exact_matcher.hh -- TEST INFRASTRUCTURE ONLY.
//
// Determinism shim for running the reference's own Stitcher::build() as a comparison target: same
// interface as PairWiseMatcher (feature/matcher.hh:40-67), answers from the reference's EXACT matcher
// FeatureMatcher::match (feature/matcher.cc:15-71) instead of the FLANN kd-forest, whose approximate
// answers change from run to run; pairs in canonical (first, second) order, because
// RANSAC's samples index the match list and the reference's own order is thread-timing dependent
// (matcher.cc:65).
#pragma once
#include <algorithm>
#include <vector>
#include "feature/matcher.hh"

namespace pano {

class ExactPairWiseMatcher {
	public:
		explicit ExactPairWiseMatcher(const std::vector<std::vector<Descriptor>>& feats): feats(feats) {}
		MatchData match(int i, int j) const {
			FeatureMatcher m(feats[i], feats[j]);
			MatchData r = m.match();
			std::sort(r.data.begin(), r.data.end());
			return r;
		}
	private:
		const std::vector<std::vector<Descriptor>>& feats;
};

}
