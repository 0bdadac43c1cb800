// The reference's matcher tests (test/test.cpp:25-56) against the drop-in headers: VLFeat descriptor dumps in,
// MatchSiftData(L2), MATLAB match indices and the 340-match ratio test.  Plain C++ (g++), no HIP headers.
// Usage: matching_dropin <vlfeat_sift1.bin> <vlfeat_sift2.bin> <match_indices1_2.bin>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cuSIFT.h"
#include "debug.h"
#include "matching.h"

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  InitCuda(0);
  int failures = 0;
  {
    // test/test.cpp:30-38: default-constructed SiftData filled by ReadVLFeatSiftData -> AddSiftData
    SiftData siftData1, siftData2;
    if (ReadVLFeatSiftData(siftData1, argv[1]) < 0 || ReadVLFeatSiftData(siftData2, argv[2]) < 0) return 2;
    const int n = ReadMATLABMatchIndices(argv[3]);
    if (n < 0) return 2;
    const uint32_t numMatches = (uint32_t)n;
    std::vector<uint32_t> indices_i(numMatches), indices_j(numMatches);
    if (ReadMATLABMatchIndices(argv[3], indices_i.data(), indices_j.data()) != n) return 2;

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
  {
    // AddSiftData (extras/debug.cpp:413-454): growth by doubling keeps host and device copies in step, and the
    // dump format round-trips (WriteVLFeatSiftData -> ReadVLFeatSiftData)
    SiftData a, b;
    if (ReadVLFeatSiftData(a, argv[1]) < 0) return 2;
    const int n1 = a.numPts, cap1 = a.maxPts;
    std::vector<SiftPoint> again(a.h_data, a.h_data + n1);
    AddSiftData(a, again.data(), n1);  // 884 + 884 > 1024 -> capacity doubles
    AddSiftData(a, again.data(), n1);  // 2652 > 2048 -> doubles again
    std::printf("AddSiftData: %d -> %d points, capacity %d -> %d\n", n1, a.numPts, cap1, a.maxPts);
    if (a.numPts != 3 * n1 || a.maxPts != 4 * cap1) ++failures;
    std::vector<SiftPoint> dev(a.numPts);
    safeCall(cusift_memcpy_d2h(cusift_dropin::ctx(), dev.data(), a.d_data, sizeof(SiftPoint) * a.numPts));
    if (std::memcmp(dev.data(), a.h_data, sizeof(SiftPoint) * a.numPts) != 0) ++failures;
    for (int k = 0; k < 3; k++)
      if (std::memcmp(a.h_data + k * n1, again.data(), sizeof(SiftPoint) * n1) != 0) ++failures;
    const char *tmp = "/tmp/cusift_dropin_roundtrip.bin";
    if (!WriteVLFeatSiftData(a, tmp) || ReadVLFeatSiftData(b, tmp) != a.numPts) ++failures;
    int same = 0;
    for (int i = 0; i < a.numPts && i < b.numPts; i++)
      same += a.h_data[i].coords2D[0] == b.h_data[i].coords2D[0] && a.h_data[i].scale == b.h_data[i].scale &&
              a.h_data[i].orientation == b.h_data[i].orientation &&
              std::memcmp(a.h_data[i].data, b.h_data[i].data, sizeof(a.h_data[i].data)) == 0;
    std::printf("dump round trip: %d / %d records identical\n", same, a.numPts);
    if (same != a.numPts) ++failures;
    std::remove(tmp);
  }
  cusift_dropin::shutdown();
  std::printf(failures ? "FAILED (%d)\n" : "PASSED\n", failures);
  return failures ? 1 : 0;
}
