// debug.h -- drop-in for the parts of the reference's extras/debug.h that move SiftData around (no OpenCV):
// AddSiftData (append with doubling), the VLFeat dump format (reader as in the reference, plus a writer so that
// extracted / gathered SiftData has a wire format), the MATLAB match-index reader and the print helpers.
// Plain C++ over the C ABI (cusift_amd.h); the cv::Mat helpers of the reference (writeMatToFile, PrintMatchData,
// ReadMATLABMatchData ...) stay out: OpenCV is a caller-side dependency.
//
// VLFeat dump (written by the reference authors' vl_sift_tofile.m, read at extras/debug.cpp:118-165):
//   uint32 numPts; float32 frames[numPts][4] = x, y, scale, orientation; float32 descriptors[numPts][128]
#ifndef CUSIFT_AMD_DEBUG_H
#define CUSIFT_AMD_DEBUG_H

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cuSIFT.h"

// extras/debug.cpp:413-454: append `numPts` host records to `data` (host and device copies), doubling the capacity
// until they fit.  Deviation: a SiftData without any buffer gets host + device buffers first -- in the reference
// such an object (the default-constructed ones of test/test.cpp:30-31) silently stores nothing.
inline void AddSiftData(SiftData &data, SiftPoint *h_data, int numPts) {
  if (numPts <= 0 || h_data == nullptr) return;
  cusift_ctx *ctx = cusift_dropin::ctx();
  if (data.h_data == nullptr && data.d_data == nullptr) {
    const int keep = data.maxPts > 0 ? data.maxPts : 1024;
    data.allocate(keep, true, true);
  }
  const int newNum = data.numPts + numPts;
  if (data.maxPts < newNum) {
    int newMax = 2 * (data.maxPts > 0 ? data.maxPts : 1);
    while (newNum > newMax) newMax *= 2;
    const size_t bytes = sizeof(SiftPoint) * (size_t)newMax, used = sizeof(SiftPoint) * (size_t)data.numPts;
    if (data.h_data != nullptr) {
      SiftPoint *grown = static_cast<SiftPoint *>(cusift_dropin::host_alloc(bytes));
      if (used) std::memcpy(grown, data.h_data, used);
      cusift_dropin::host_free(data.h_data);
      data.h_data = grown;
    }
    if (data.d_data != nullptr) {
      void *grown = nullptr;
      safeCall(cusift_malloc(&grown, bytes));
      if (used) safeCall(cusift_memcpy_d2d(ctx, grown, data.d_data, used));
      cusift_free(data.d_data);
      data.d_data = static_cast<SiftPoint *>(grown);
    }
    data.maxPts = newMax;
  }
  const size_t add = sizeof(SiftPoint) * (size_t)numPts;
  if (data.h_data != nullptr) std::memcpy(data.h_data + data.numPts, h_data, add);
  if (data.d_data != nullptr) safeCall(cusift_memcpy_h2d(ctx, data.d_data + data.numPts, h_data, add));
  data.numPts = newNum;
}

// extras/debug.cpp:118-165.  Frames go to coords2D / scale / orientation (radians, as VLFeat wrote them), the
// descriptor to data[]; every other field of the new records is zero.  Returns the number of points read
// (-1: cannot open / truncated file; the reference does not check).
inline int ReadVLFeatSiftData(SiftData &siftData, const char *filename) {
  std::fprintf(stderr, "Reading vlfeat data from %s", filename);
  FILE *fp = std::fopen(filename, "rb");
  if (!fp) {
    std::fprintf(stderr, " ... cannot open\n");
    return -1;
  }
  uint32_t n = 0;
  bool ok = std::fread(&n, sizeof(n), 1, fp) == 1;
  std::vector<float> frames, desc;
  if (ok) {
    frames.resize(4 * (size_t)n);
    desc.resize(128 * (size_t)n);
    ok = std::fread(frames.data(), sizeof(float), frames.size(), fp) == frames.size() &&
         std::fread(desc.data(), sizeof(float), desc.size(), fp) == desc.size();
  }
  std::fclose(fp);
  if (!ok) {
    std::fprintf(stderr, " ... truncated\n");
    return -1;
  }
  std::fprintf(stderr, " ... and got %d points\n", (int)n);
  std::vector<SiftPoint> recs(n);
  if (n) std::memset(recs.data(), 0, sizeof(SiftPoint) * n);
  for (uint32_t i = 0; i < n; ++i) {
    recs[i].coords2D[0] = frames[4 * (size_t)i];
    recs[i].coords2D[1] = frames[4 * (size_t)i + 1];
    recs[i].scale = frames[4 * (size_t)i + 2];
    recs[i].orientation = frames[4 * (size_t)i + 3];
    std::memcpy(recs[i].data, &desc[128 * (size_t)i], sizeof(float) * 128);
  }
  AddSiftData(siftData, recs.data(), (int)n);
  return (int)n;
}

// The same format, written from a SiftData's host records (call Synchronize() first if only the device copy is
// current).  New: the reference only reads this format; test/detector.cpp:52-63 writes the 4-column variant
// without descriptors.  Orientation is stored as it is in the records (degrees for extracted SiftData).
inline bool WriteVLFeatSiftData(const SiftData &siftData, const char *filename) {
  if (siftData.h_data == nullptr && siftData.numPts > 0) return false;
  FILE *fp = std::fopen(filename, "wb");
  if (!fp) return false;
  const uint32_t n = (uint32_t)(siftData.numPts > 0 ? siftData.numPts : 0);
  bool ok = std::fwrite(&n, sizeof(n), 1, fp) == 1;
  for (uint32_t i = 0; ok && i < n; ++i) {
    const SiftPoint &p = siftData.h_data[i];
    const float frame[4] = {p.coords2D[0], p.coords2D[1], p.scale, p.orientation};
    ok = std::fwrite(frame, sizeof(float), 4, fp) == 4;
  }
  for (uint32_t i = 0; ok && i < n; ++i) ok = std::fwrite(siftData.h_data[i].data, sizeof(float), 128, fp) == 128;
  return (std::fclose(fp) == 0) && ok;
}

// extras/debug.cpp:167-181: uint32 n; uint32 i[n]; uint32 j[n] (1-based MATLAB indices).  Returns n.
inline int ReadMATLABMatchIndices(const char *indices_filename, uint32_t *indices_i = nullptr,
                                  uint32_t *indices_j = nullptr) {
  std::fprintf(stderr, "Reading match indices data from %s\n", indices_filename);
  FILE *fp = std::fopen(indices_filename, "rb");
  if (!fp) return -1;
  uint32_t n = 0;
  bool ok = std::fread(&n, sizeof(n), 1, fp) == 1;
  if (ok && indices_i != nullptr && indices_j != nullptr)
    ok = std::fread(indices_i, sizeof(uint32_t), n, fp) == n && std::fread(indices_j, sizeof(uint32_t), n, fp) == n;
  std::fclose(fp);
  return ok ? (int)n : -1;
}

// extras/debug.cpp:26-73: one block of text per keypoint; a SiftData without host records gets them first.
inline void PrintSiftData(SiftData &data) {
  if (data.h_data == nullptr && data.d_data != nullptr && data.maxPts > 0) {
    data.h_data = static_cast<SiftPoint *>(cusift_dropin::host_alloc(sizeof(SiftPoint) * (size_t)data.maxPts));
    data.Synchronize();
  }
  const SiftPoint *h = data.h_data;
  for (int i = 0; h != nullptr && i < data.numPts; ++i) {
    std::printf("xpos         = %.2f\n", h[i].coords2D[0]);
    std::printf("ypos         = %.2f\n", h[i].coords2D[1]);
    std::printf("scale        = %.2f\n", h[i].scale);
    std::printf("sharpness    = %.2f\n", h[i].sharpness);
    std::printf("edgeness     = %.2f\n", h[i].edgeness);
    std::printf("orientation  = %.2f\n", h[i].orientation);
    std::printf("score        = %.2f\n", h[i].score);
    for (int j = 0; j < 8; ++j) {
      std::printf(j == 0 ? "data = " : "       ");
      for (int k = 0; k < 16; ++k) {
        const float v = h[i].data[j * 16 + k];
        if (v < 0.01f)
          std::printf(" .   ");
        else
          std::printf("%.2f ", v);
      }
      std::printf("\n");
    }
  }
  std::printf("Number of available points: %d\n", data.numPts);
  std::printf("Number of allocated points: %d\n", data.maxPts);
}

// extras/debug.cpp:95-115: tab-separated x, y, match x, match y and the two linear pixel indices per keypoint.
inline bool PrintMatchSiftData(SiftData &siftData1, const char *filename, int imgw) {
  FILE *fp = std::fopen(filename, "w");
  if (!fp) {
    std::printf("File Not Opened\n");
    return false;
  }
  const SiftPoint *s = siftData1.h_data;
  for (int i = 0; s != nullptr && i < siftData1.numPts; ++i) {
    const int ind = (int)s[i].coords2D[0] + (int)s[i].coords2D[1] * imgw;
    const int ind2 = (int)s[i].match_xpos + (int)s[i].match_ypos * imgw;
    std::fprintf(fp, "%g\t%g\t%g\t%g\t%d\t%d\t\n", s[i].coords2D[0], s[i].coords2D[1], s[i].match_xpos,
                 s[i].match_ypos, ind, ind2);
  }
  return std::fclose(fp) == 0;
}

#endif  // CUSIFT_AMD_DEBUG_H
