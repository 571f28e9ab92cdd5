#!/usr/bin/env python3
"""INTEGRATION.md's hooks as an executable recipe -- TEST INFRASTRUCTURE (build time only).

Reads the reference's orchestration sources where they lie (never copied into this repository):
    src/stitch/{stitcherbase.hh, stitcherbase.cc, stitcher.hh, stitcher.cc, cylstitcher.hh, cylstitcher.cc}
and writes two edited variants under the output directory (oracle/_ref/variants, git-ignored):

  exact/   the reference's Stitcher / CylinderStitcher with ONE change: PairWiseMatcher (FLANN kd-forest,
           approximate and run-to-run non-deterministic: SURVEY F2/F3) -> ExactPairWiseMatcher, a thin wrapper
           of the reference's own exact FeatureMatcher (oracle/ref_exact_matcher.hh).  With the mt19937 seed
           seam of ref_driver.cc this is the deterministic CPU reference build() is compared against.
  hip/     the same files with the five construction-site edits of INTEGRATION.md:
             1  new SIFTDetector                    -> new HipSIFTDetector                 (stitcherbase.hh)
             2  PairWiseMatcher                     -> HipPairWiseMatcher                  (stitcher.*, cylstitcher.cc)
             3  TransformEstimation(...)            -> HipTransformEstimation(...)         (stitcher.cc, cylstitcher.cc)
             4  bundle.blend()                      -> hip_blend(bundle)                   (stitcher.cc, cylstitcher.cc)
             5  CylinderWarper warper(...)          -> HipCylinderWarper warper(...)       (cylstitcher.cc)
           hipfast/ additionally carries INTEGRATION.md's batched forms (1b, 2b, 3b) and the optional HOST hook
             6  CameraEstimator{...}.estimate()     -> HostCameraEstimator{...}.estimate() (stitcher.cc)

So that both variants and the untouched reference classes can live in ONE test process, the three classes are
renamed per variant (Stitcher -> ExactStitcher / HookedStitcher, ...).  A maintainer applying the hooks to the
real tree makes only the five edits above.
Usage: apply_hooks.py <reference root> <output dir>
"""
import os
import re
import sys

FILES = ["stitcherbase.hh", "stitcherbase.cc", "stitcher.hh", "stitcher.cc", "cylstitcher.hh", "cylstitcher.cc"]


def rename_classes(text, prefix):
    text = re.sub(r"\bStitcherBase\b", prefix + "StitcherBase", text)
    text = re.sub(r"\bCylinderStitcher\b", prefix + "CylinderStitcher", text)
    text = re.sub(r"\bStitcher\b", prefix + "Stitcher", text)
    return text


def must_sub(pattern, repl, text, name, count_min=1):
    new, n = re.subn(pattern, repl, text, flags=re.S)
    if n < count_min:
        raise SystemExit(f"apply_hooks: pattern {pattern!r} matched {n} time(s) in {name} (expected >= {count_min}): the reference changed")
    return new


def variant_exact(name, text, rename=True):
    if rename:
        text = rename_classes(text, "Exact")
    if name in ("stitcher.hh", "stitcher.cc", "cylstitcher.cc"):
        text = must_sub(r"\bPairWiseMatcher\b", "ExactPairWiseMatcher", text, name)
    if name == "stitcherbase.hh":
        text = must_sub(r'#include "feature/feature.hh"', '#include "feature/feature.hh"\n#include "ref_exact_matcher.hh"', text, name)
    return text


def variant_hip(name, text, prefix="Hooked", fast=False):
    """the five construction-site hooks; fast=True: INTEGRATION.md's batched forms on top (hooks 1b, 2b, 3b)"""
    if prefix:
        text = rename_classes(text, prefix)
    base = (prefix or "") + "StitcherBase"
    if name == "stitcherbase.hh":
        text = must_sub(r'#include "feature/feature.hh"', '#include "feature/feature.hh"\n#include "pano_hip.hh"', text, name)
        text = must_sub(r"new SIFTDetector\b", "new HipSIFTDetector", text, name)                         # hook 1
        if fast:                                                                                          # hook 1b: the resident feature set
            text = must_sub(r"(std::vector<std::vector<Descriptor>> feats;)", r"\1\n\t\tHipFeatureSet hip_feats;\t// the same features, resident in HBM", text, name)
    if name == "stitcherbase.cc" and fast:
        # hook 1b: body of calc_feature() (stitcherbase.cc:9-27) = one batched device call for all images
        body = (
            "#pragma omp parallel for schedule(dynamic)\n"
            "  REP(k, (int)imgs.size()) imgs[k].load();          // decode in parallel, as the reference's loop does\n"
            "  std::vector<const Mat32f*> ptrs;\n"
            "  for (auto& r : imgs) ptrs.push_back(r.img);\n"
            "  hip_feats = static_cast<HipSIFTDetector&>(*feature_det).calc_feature(ptrs);\n"
            "  feats = hip_feats.feats;\n"
            "  REP(k, (int)imgs.size()) {\n"
            "    if (config::LAZY_READ) imgs[k].release();\n"
            "    print_debug(\"Image %d has %lu features\\n\", k, feats[k].size());\n"
            "    keypoints[k].resize(feats[k].size());\n"
            "    REP(i, feats[k].size()) keypoints[k][i] = feats[k][i].coor;\n"
            "  }\n}\n")
        text = must_sub(r"#pragma omp parallel for schedule\(dynamic\)\s*\n\s*REP\(k, \(int\)imgs\.size\(\)\) \{.*?\n  \}\n\}\n", lambda m: body, text, name)
    if name in ("stitcher.hh", "stitcher.cc"):
        if fast:                                                                                          # hook 2b
            text = must_sub(r"\bPairWiseMatcher pwmatcher\(feats\)", "HipBatchedMatcher pwmatcher(hip_feats, imgs)", text, name, 2 if name == "stitcher.cc" else 0)
            text = must_sub(r"\bPairWiseMatcher\b", "HipBatchedMatcher", text, name)
        else:
            text = must_sub(r"\bPairWiseMatcher\b", "HipPairWiseMatcher", text, name)                     # hook 2
    if name == "cylstitcher.cc":
        if fast:                                                                                          # descriptors already resident: no second H2D
            text = must_sub(r"\bPairWiseMatcher pwmatcher\(feats\)", "HipPairWiseMatcher pwmatcher(hip_feats)", text, name)
        else:
            text = must_sub(r"\bPairWiseMatcher\b", "HipPairWiseMatcher", text, name)
    if name in ("stitcher.cc", "cylstitcher.cc"):
        te = "HipBatchedTransformEstimation" if (fast and name == "stitcher.cc") else "HipTransformEstimation"   # hook 3 / 3b
        text = must_sub(r"(?<![\w:])TransformEstimation(\s*\(|\s+transf\s*\()", te + r"\1", text, name)
        text = must_sub(r"\bbundle\.blend\(\)", "hip_blend(bundle)", text, name)                            # hook 4
    if name == "stitcher.cc" and fast:
        text = must_sub(r"\bCameraEstimator\{", "HostCameraEstimator{", text, name)                        # hook 6 (host-only, optional)
        # hook 2b, second half: with every pair already matched and estimated by the matcher object, the bodies of the two pair
        # loops (stitcher.cc:106-109, :120-134) are table lookups, and run in parallel they do nothing but race in the reference's
        # print_debug (lib/debugutils.cc:33-38 inserts into a static std::map outside its critical section; under -DDEBUG 703
        # near-simultaneous calls corrupt it -- seen as a segfault in match_image and as free() aborts at exit): serial loops
        text = must_sub(r"#pragma omp parallel for schedule\(dynamic\)\s*\n(\s*REP\((?:k, \(int\)tasks\.size\(\)|i, n)\))", r"\1", text, name, 2)
    if name == "cylstitcher.cc":
        text = must_sub(r"\bCylinderWarper warper\b", "HipCylinderWarper warper", text, name, 2)            # hook 5
    return text


VARIANTS = {
    # one test process holds exact + hip + hipfast next to the untouched classes: renamed
    "exact": lambda n, t: variant_exact(n, t),
    "hip": lambda n, t: variant_hip(n, t, "Hooked", False),
    "hipfast": lambda n, t: variant_hip(n, t, "Batched", True),
    # what a maintainer's tree looks like (no renames): the reference's main.cc is compiled against these
    "cli_exact": lambda n, t: variant_exact(n, t, rename=False),
    "cli_hip": lambda n, t: variant_hip(n, t, None, False),
    "cli_hipfast": lambda n, t: variant_hip(n, t, None, True),
}


def main():
    ref, out = sys.argv[1], sys.argv[2]
    for var, fn in VARIANTS.items():
        # cli_* variants mirror the tree layout (stitch/...) so that main.cc's #include "stitch/stitcher.hh" finds them
        d = os.path.join(out, var, "stitch") if var.startswith("cli_") else os.path.join(out, var)
        os.makedirs(d, exist_ok=True)
        for name in FILES:
            text = open(os.path.join(ref, "src", "stitch", name)).read()
            with open(os.path.join(d, name), "w") as f:
                f.write(f"// GENERATED by oracle/apply_hooks.py ({var}) from the reference's src/stitch/{name} -- build output, never committed\n")
                f.write(fn(name, text))
        if var.startswith("cli_"):
            # main.cc itself is not edited, but it must be COMPILED from a place where "stitch/stitcher.hh" means the hooked
            # header: a quoted include looks in the including file's own directory before any -I path, and main.cc
            # instantiates the StitcherBase constructor template (which detector is built, how the object is laid out)
            with open(os.path.join(out, var, "main.cc"), "w") as f:
                f.write(f"// GENERATED by oracle/apply_hooks.py ({var}): the reference's src/main.cc, unedited -- build output, never committed\n")
                f.write(open(os.path.join(ref, "src", "main.cc")).read())
    print(f"apply_hooks: wrote {', '.join(VARIANTS)} under {out}")


if __name__ == "__main__":
    main()
