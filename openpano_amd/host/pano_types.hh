// pano_types.hh -- standalone value types of the C++ host mirror (namespace pano / config).
//
// Used when the adapters in pano_hip.hh are built WITHOUT the reference tree.  Names, members
// and meanings follow the reference so that host code written against it compiles unchanged:
//   Vec2D / Coor / Vec             lib/geometry.hh:33-246
//   Mat<T>, Mat32f                 lib/mat.h:8-57   (row-major H x W x C, shared ownership)
//   Descriptor                     feature/feature.hh:18-31
//   MatchData                      feature/matcher.hh:14-25
//   Shape2D, MatchInfo             stitch/match_info.hh:14-78
//   Homography                     stitch/homography.hh:20-140 (storage + trans; inverse() via the C-ABI helper)
//   ImageRef                       stitch/imageref.hh:13-40 (in-memory: the CLI's file loading is out of scope)
//   namespace config               lib/config.hh:24-85 with the shipped defaults of src/config.cfg
// With -DOPENPANO_WITH_REFERENCE pano_hip.hh includes the reference's own headers instead and this
// file is not used.  Written from the interface description in SURVEY.md; no reference code.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <istream>
#include <ostream>
#include <limits>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "pano_la.hh"

namespace config {
inline bool CYLINDER = false, TRANS = false, CROP = true, ESTIMATE_CAMERA = true, STRAIGHTEN = true;
inline bool ORDERED_INPUT = false, LAZY_READ = true;
inline float FOCAL_LENGTH = 37.f;
inline int MAX_OUTPUT_SIZE = 8000;
inline int SIFT_WORKING_SIZE = 800, NUM_OCTAVE = 4, NUM_SCALE = 7;
inline float SCALE_FACTOR = 1.4142135623f, GAUSS_SIGMA = 1.4142135623f;
inline int GAUSS_WINDOW_FACTOR = 6;
inline float JUDGE_EXTREMA_DIFF_THRES = 2e-3f, CONTRAST_THRES = 4e-2f, PRE_COLOR_THRES = 5e-2f, EDGE_RATIO = 6.f;
inline int CALC_OFFSET_DEPTH = 4;
inline float OFFSET_THRES = 0.5f, ORI_RADIUS = 4.5f;
inline int ORI_HIST_SMOOTH_COUNT = 2, DESC_HIST_SCALE_FACTOR = 3, DESC_INT_FACTOR = 512;
inline float MATCH_REJECT_NEXT_RATIO = 0.8f;
inline int RANSAC_ITERATIONS = 1500;
inline double RANSAC_INLIER_THRES = (double)3.5f;
inline float INLIER_IN_MATCH_RATIO = 0.1f, INLIER_IN_POINTS_RATIO = 0.04f;
inline int MULTIBAND = 0;
inline int MULTIPASS_BA = 1;
inline float LM_LAMBDA = 5.f;
}	// namespace config

template <typename T>
struct Vector2D {
	T x = 0, y = 0;
	Vector2D() {}
	explicit Vector2D(T mx, T my): x(mx), y(my) {}
	Vector2D operator+(const Vector2D& v) const { return Vector2D(x + v.x, y + v.y); }
	Vector2D operator-(const Vector2D& v) const { return Vector2D(x - v.x, y - v.y); }
	Vector2D operator*(T f) const { return Vector2D(x * f, y * f); }
	T dot(const Vector2D& v) const { return x * v.x + y * v.y; }
	bool isNaN() const { return std::isnan((double)x); }
};
template <typename T>
struct Vector {
	T x = 0, y = 0, z = 0;
	explicit Vector(T mx = 0, T my = 0, T mz = 0): x(mx), y(my), z(mz) {}
	// lib/geometry.hh:60-70
	T sqr() const { return x * x + y * y + z * z; }
	T dot(const Vector& v) const { return x * v.x + y * v.y + z * v.z; }
	Vector cross(const Vector& v) const { return Vector(y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x); }
	Vector operator*(T f) const { return Vector(x * f, y * f, z * f); }
};
typedef Vector<double> Vec;
typedef Vector2D<int> Coor;
typedef Vector2D<double> Vec2D;

template <typename T>
class Mat {
	public:
		Mat() {}
		Mat(int rows, int cols, int channels):
			m_rows(rows), m_cols(cols), m_channels(channels),
			m_data{new T[(size_t)rows * cols * channels], std::default_delete<T[]>()} {}
		T& at(int r, int c, int ch = 0) { return ptr(r)[c * m_channels + ch]; }
		const T& at(int r, int c, int ch = 0) const { return ptr(r)[c * m_channels + ch]; }
		Mat<T> clone() const {
			Mat<T> res(m_rows, m_cols, m_channels);
			memcpy(res.ptr(0), ptr(0), sizeof(T) * (size_t)m_rows * m_cols * m_channels);
			return res;
		}
		const T* ptr(int r = 0) const { return m_data.get() + (size_t)r * m_cols * m_channels; }
		T* ptr(int r = 0) { return m_data.get() + (size_t)r * m_cols * m_channels; }
		int height() const { return m_rows; }
		int width() const { return m_cols; }
		int rows() const { return m_rows; }
		int cols() const { return m_cols; }
		int channels() const { return m_channels; }
		int pixels() const { return m_rows * m_cols; }
	protected:
		int m_rows = 0, m_cols = 0, m_channels = 0;
		std::shared_ptr<T> m_data;
};
using Mat32f = Mat<float>;

namespace pano {

struct Descriptor {
	Vec2D coor;
	std::vector<float> descriptor;
};

class MatchData {
	public:
		std::vector<std::pair<int, int>> data;     // <idx in first, idx in second>
		int size() const { return (int)data.size(); }
		void reverse() { for (auto& i : data) i = std::make_pair(i.second, i.first); }
};

struct Shape2D {
	int w, h;
	Shape2D(int w, int h): w(w), h(h) {}
	double halfw() const { return w * 0.5; }
	double halfh() const { return h * 0.5; }
	Vec2D center() const { return Vec2D{halfw(), halfh()}; }
};

class Homography {
	public:
		double data[9];
		Homography() {}
		Homography(const double (&arr)[9]) { for (int i = 0; i < 9; ++i) data[i] = arr[i]; }
		double& operator[](int idx) { return data[idx]; }
		const double& operator[](int idx) const { return data[idx]; }
		Vec trans(const Vec& m) const {
			return Vec(data[0] * m.x + data[1] * m.y + data[2] * m.z,
					data[3] * m.x + data[4] * m.y + data[5] * m.z,
					data[6] * m.x + data[7] * m.y + data[8] * m.z);
		}
		Vec2D trans2d(const Vec2D& m) const {
			Vec r = trans(Vec(m.x, m.y, 1));
			double denom = 1.0 / r.z;
			return Vec2D(r.x * denom, r.y * denom);
		}
		static Homography I() { Homography r; for (int i = 0; i < 9; ++i) r.data[i] = (i % 4 == 0); return r; }
		Vec trans(const Vec2D& m) const { return trans(Vec(m.x, m.y, 1)); }
		// stitch/homography.hh:36-50, homography.cc:25-48 (the product is a plain row-times-column sum
		// from 0; the inverse is Eigen's FullPivLU in the reference, pano_la::inverse3 here)
		Homography transpose() const {
			const double t[9] = {data[0], data[3], data[6], data[1], data[4], data[7], data[2], data[5], data[8]};
			return Homography(t);
		}
		Homography operator*(const Homography& r) const {
			Homography ret;
			for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
				double s = 0;
				for (int k = 0; k < 3; ++k) s += data[i * 3 + k] * r.data[k * 3 + j];
				ret.data[i * 3 + j] = s;
			}
			return ret;
		}
		void operator+=(const Homography& r) { for (int i = 0; i < 9; ++i) data[i] += r.data[i]; }
		void mult(double r) { for (int i = 0; i < 9; ++i) data[i] *= r; }
		Homography inverse(bool* succ = nullptr) const {
			Homography ret;
			const bool ok = pano_la::inverse3(data, ret.data);
			if (succ) *succ = ok;
			else if (!ok) { fprintf(stderr, "Homography::inverse: singular matrix\n"); abort(); }   // m_assert(lu.isInvertible())
			return ret;
		}
		static Homography get_translation(double dx, double dy) { const double t[9] = {1, 0, dx, 0, 1, dy, 0, 0, 1}; return Homography(t); }
		// text form of stitch/homography.hh:154-163 (default stream precision, like the reference)
		void serialize(std::ostream& os) const { for (int i = 0; i < 8; ++i) os << data[i] << " "; os << data[8]; }
		static Homography deserialize(std::istream& is) { Homography r; for (int i = 0; i < 9; ++i) is >> r[i]; return r; }
};

struct MatchInfo {
	typedef std::pair<Vec2D, Vec2D> PCC;           // to, from (half-shifted coordinates)
	std::vector<PCC> match;
	float confidence = 0;                          // negative: -#inliers of a rejected pair
	Homography homo;
	void reverse() { for (auto& c : match) std::swap(c.first, c.second); }
	// stitch/match_info.hh:26-50: "confidence homo[9] n  to.x to.y from.x from.y ..."
	void serialize(std::ostream& os) const {
		os << confidence << " ";
		homo.serialize(os);
		os << " " << match.size();
		for (auto& p : match) os << " " << p.first.x << " " << p.first.y << " " << p.second.x << " " << p.second.y;
	}
	static MatchInfo deserialize(std::istream& is) {
		MatchInfo ret;
		is >> ret.confidence;
		ret.homo = Homography::deserialize(is);
		int match_size;
		is >> match_size;
		ret.match.resize(match_size);
		for (int i = 0; i < match_size; ++i) { PCC& p = ret.match[i]; is >> p.first.x >> p.first.y >> p.second.x >> p.second.y; }
		return ret;
	}
};

// Stitcher::dump_matchinfo / load_matchinfo (stitch/debug.cc:111-140): the reference's on-disk
// checkpoint of the match + RANSAC stage ("i j" line, then the serialized MatchInfo; only pairs
// with positive confidence), usable as a cross-implementation fixture
inline void dump_matchinfo(const char* fname, const std::vector<std::vector<MatchInfo>>& pairwise_matches) {
	std::ofstream fout(fname);
	const int n = (int)pairwise_matches.size();
	for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
		const MatchInfo& m = pairwise_matches[i][j];
		if (m.confidence <= 0) continue;
		fout << i << " " << j << std::endl;
		m.serialize(fout);
		fout << std::endl;
	}
}
inline std::vector<std::vector<MatchInfo>> load_matchinfo(const char* fname, int n) {
	std::vector<std::vector<MatchInfo>> pm(n, std::vector<MatchInfo>(n));
	std::ifstream fin(fname);
	int i, j;
	while (true) {
		fin >> i >> j;
		if (fin.eof() || !fin) break;
		pm[i][j] = MatchInfo::deserialize(fin);
	}
	return pm;
}

// in-memory image reference (the reference's lazy file loading stays with its CLI)
struct ImageRef {
	std::string fname;
	Mat32f* img = nullptr;
	int _width = 0, _height = 0;
	ImageRef(const std::string& fname): fname(fname) {}
	explicit ImageRef(const Mat32f& m): fname("<memory>"), img(new Mat32f(m)), _width(m.width()), _height(m.height()) {}
	ImageRef(const ImageRef& o): fname(o.fname), img(o.img ? new Mat32f(*o.img) : nullptr), _width(o._width), _height(o._height) {}
	ImageRef& operator=(const ImageRef&) = delete;
	~ImageRef() { delete img; }
	void load() {}
	void release() {}
	int width() const { return _width; }
	int height() const { return _height; }
	Shape2D shape() const { return {_width, _height}; }
};

}	// namespace pano
