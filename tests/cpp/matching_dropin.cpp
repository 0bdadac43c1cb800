// The reference's matcher tests (test/test.cpp:25-56) against the drop-in headers: VLFeat descriptor dumps in,
// MatchSiftData(L2), MATLAB match indices and the 340-match ratio test.  Plain C++ (g++), no HIP headers.
// Usage: matching_dropin <vlfeat_sift1.bin> <vlfeat_sift2.bin> <match_indices1_2.bin>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cuSIFT.h"
#include "matching.h"

// extras/debug.cpp:118-165 (ReadVLFeatSiftData) + :413-454 (AddSiftData): host records -> SiftData (host + device)
static bool read_vlfeat(SiftData &data, const char *path) {
  FILE *fp = std::fopen(path, "rb");
  if (!fp) return false;
  uint32_t n = 0;
  if (std::fread(&n, sizeof(n), 1, fp) != 1) return false;
  std::vector<float> pts(4 * (size_t)n), desc(128 * (size_t)n);
  if (std::fread(pts.data(), sizeof(float), pts.size(), fp) != pts.size()) return false;
  if (std::fread(desc.data(), sizeof(float), desc.size(), fp) != desc.size()) return false;
  std::fclose(fp);
  InitSiftData(data, (int)n, true, true);
  std::memset(data.h_data, 0, sizeof(SiftPoint) * n);
  for (uint32_t i = 0; i < n; i++) {
    data.h_data[i].coords2D[0] = pts[4 * i];
    data.h_data[i].coords2D[1] = pts[4 * i + 1];
    data.h_data[i].scale = pts[4 * i + 2];
    data.h_data[i].orientation = pts[4 * i + 3];
    std::memcpy(data.h_data[i].data, &desc[128 * (size_t)i], sizeof(float) * 128);
  }
  data.numPts = (int)n;
  safeCall(cusift_memcpy_h2d(cusift_dropin::ctx(), data.d_data, data.h_data, sizeof(SiftPoint) * n));
  return true;
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  InitCuda(0);
  int failures = 0;
  {
    SiftData siftData1, siftData2;
    if (!read_vlfeat(siftData1, argv[1]) || !read_vlfeat(siftData2, argv[2])) return 2;
    FILE *fp = std::fopen(argv[3], "rb");
    if (!fp) return 2;
    uint32_t numMatches = 0;
    if (std::fread(&numMatches, sizeof(uint32_t), 1, fp) != 1) return 2;
    std::vector<uint32_t> indices_i(numMatches), indices_j(numMatches);
    if (std::fread(indices_i.data(), sizeof(uint32_t), numMatches, fp) != numMatches) return 2;
    if (std::fread(indices_j.data(), sizeof(uint32_t), numMatches, fp) != numMatches) return 2;
    std::fclose(fp);

    // TEST(Matching, MatchingTest)
    std::vector<SiftMatch *> matches = MatchSiftData(siftData1, siftData2, MatchSiftDistanceL2);
    int agree = 0;
    for (uint32_t i = 0; i < numMatches; i++)
      agree += (int)indices_j[i] == matches[indices_i[i] - 1]->pt1->match + 1;
    std::printf("MATLAB match indices: %d / %u agree; %zu matches with default thresholds\n", agree, numMatches,
                matches.size());
    if (agree != (int)numMatches || (int)matches.size() != siftData1.numPts) ++failures;
    for (SiftMatch *m : matches) delete m;

    // TEST(Matching, MatchingRatioTest)
    matches = MatchSiftData(siftData1, siftData2, MatchSiftDistanceL2, 1000, 0.6);
    std::printf("ratio test: %zu matches (reference expects 340)\n", matches.size());
    if (matches.size() != 340) ++failures;
    for (SiftMatch *m : matches) delete m;
  }
  cusift_dropin::shutdown();
  std::printf(failures ? "FAILED (%d)\n" : "PASSED\n", failures);
  return failures ? 1 : 0;
}
