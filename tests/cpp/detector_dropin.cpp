// The reference's own hot-path test (test/detector.cpp:18-90) re-written against the drop-in header:
// same calls, same parameters, same golden file -- minus OpenCV/gtest (the image comes from the PGM fixture).
// Plain C++ (g++), no HIP headers.  Usage: detector_dropin <gray1.pgm> <cusift1_check.bin>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cuImage.h"
#include "cuSIFT.h"

static bool read_pgm(const char *path, std::vector<float> &img, int &w, int &h) {
  FILE *fp = std::fopen(path, "rb");
  if (!fp) return false;
  int maxv = 0;
  if (std::fscanf(fp, "P5 %d %d %d", &w, &h, &maxv) != 3 || maxv != 255) return false;
  std::fgetc(fp);
  std::vector<unsigned char> raw((size_t)w * h);
  if (std::fread(raw.data(), 1, raw.size(), fp) != raw.size()) return false;
  std::fclose(fp);
  img.assign(raw.begin(), raw.end());
  return true;
}

int main(int argc, char **argv) {
  if (argc < 3) {
    std::printf("usage: %s gray1.pgm cusift1_check.bin\n", argv[0]);
    return 2;
  }
  std::vector<float> im;
  int w = 0, h = 0;
  if (!read_pgm(argv[1], im, w, h)) return 2;

  if (!deviceInit(0)) return 2;  // cutils.h:50-69 (calls InitCuda)
  cuImage *cuIm = new cuImage(w, h, im.data());
  safeThreadSync();                // cutils.h:21,32-39
  checkMsg("upload");              // cutils.h:22,41-48 (compiles, no sticky error to poll: see cuSIFT.h)

  // ---- live API, test/detector.cpp:37-49 ----
  SiftData *siftData = new SiftData(4096, true, true);
  siftData->numOctaves = 6;
  siftData->initBlur = 0.0f;
  siftData->peakThresh = 0.1f;
  siftData->edgeThresh = 10.0f;
  siftData->lowestScale = 0.0f;
  siftData->initSubsampling = 1.0f;
  TimerCPU wall(0.0f);
  TimerGPU gpu(0);  // cutils.h:94-114: events on the stream, as the reference brackets its stages (cuSIFT.cu:64)
  siftData->Extract(im.data(), w, h);
  const float gpu_ms = gpu.read(), wall_ms = wall.read();
  std::printf("Extract: TimerGPU %.3f ms inside TimerCPU %.3f ms\n", gpu_ms, wall_ms);
  if (!(gpu_ms > 0.0f && gpu_ms <= wall_ms + 0.05f)) {
    std::printf("TimerGPU out of range\n");
    return 3;
  }

  FILE *fp = std::fopen(argv[2], "rb");
  if (!fp) return 2;
  uint32_t numPts = 0;
  if (std::fread(&numPts, sizeof(uint32_t), 1, fp) != 1) return 2;
  std::printf("num pts: golden %u, extracted %d\n", numPts, siftData->numPts);
  int failures = 0;
  if ((int)numPts != siftData->numPts) ++failures;  // ASSERT_EQ(numPts, siftData->numPts)

  // the reference's one-sided "really hacky" check (test/detector.cpp:71-84) and a two-sided one on the
  // 1555 coarse-octave rows that the saturated golden run determines uniquely
  int hacky_found = 0, strict_found = 0;
  std::vector<float> gold((size_t)numPts * 4);
  if (std::fread(gold.data(), sizeof(float), gold.size(), fp) != gold.size()) return 2;
  std::fclose(fp);
  for (uint32_t i = 0; i < numPts; i++) {
    const float *d = &gold[4 * i];
    bool found = false, strict = false;
    for (int j = 0; j < siftData->numPts; j++) {
      const SiftPoint &pt = siftData->h_data[j];
      if (pt.coords2D[0] - d[0] < 0.1 && pt.coords2D[1] - d[1] < 0.1 && pt.scale - d[2] < 0.1 && pt.orientation - d[3] < 0.1)
        found = true;
      if (std::fabs(pt.coords2D[0] - d[0]) < 1e-2 && std::fabs(pt.coords2D[1] - d[1]) < 1e-2 && std::fabs(pt.scale - d[2]) < 1e-2)
        strict = true;
    }
    hacky_found += found;
    if (i < 1555) strict_found += strict;
  }
  std::printf("hacky check: %d / %u found; strict coarse-octave check: %d / 1555 within 1e-2\n", hacky_found, numPts,
              strict_found);
  // The hacky one-sided check depends on WHICH octave-0 points survive the 4096 cap (racy in the reference,
  // SURVEY section 4): it is reported, not asserted.  Asserted instead: the coarse rows above, and below that
  // every golden row is one of the 9508 points an unsaturated run finds.
  if (strict_found != 1555) ++failures;
  {
    SiftData full(16384, true, true);
    full.numOctaves = 6;
    full.initBlur = 0.0f;
    full.peakThresh = 0.1f;
    full.edgeThresh = 10.0f;
    full.lowestScale = 0.0f;
    full.Extract(im.data(), w, h);
    int found_all = 0, ori_ok = 0;
    for (uint32_t i = 0; i < numPts; i++) {
      const float *d = &gold[4 * i];
      for (int j = 0; j < full.numPts; j++) {
        const SiftPoint &pt = full.h_data[j];
        if (std::fabs(pt.coords2D[0] - d[0]) < 1e-2 && std::fabs(pt.coords2D[1] - d[1]) < 1e-2 &&
            std::fabs(pt.scale - d[2]) < 1e-2) {
          ++found_all;
          float da = std::fabs(pt.orientation - d[3]);
          if (da > 180.0f) da = 360.0f - da;
          ori_ok += da < 1e-3f;
          break;
        }
      }
    }
    std::printf("unsaturated run: %d points; golden rows found within 1e-2: %d / %u; of those, orientation within 1e-3 degree: %d\n",
                full.numPts, found_all, numPts, ori_ok);
    if (full.numPts != 9508) ++failures;
    if (found_all < (int)numPts - 2) ++failures;
    // the reference's own orientations (test/detector.cpp:79 compares them to 0.1, one-sided): >= 98 % within 1e-3 degree
    if (ori_ok < (int)(0.98 * found_all)) ++failures;
  }

  // ---- legacy API, main.cpp:313-328,348-349 ----
  cuImage img1;
  img1.Allocate(w, h, iAlignUp(w, 128), false, NULL, im.data());
  img1.HostToDevice();
  SiftData siftData1;
  InitSiftData(siftData1, 4096, true, true);
  ExtractSift(siftData1, img1, 6, 0.0f, 0.1f, 0.0f);
  std::printf("legacy ExtractSift: %d points\n", siftData1.numPts);
  if (siftData1.numPts != siftData->numPts) ++failures;
  // the coarse-octave block (first 1555 records) is deterministic: same set from both entry points
  double acc0 = 0, acc1 = 0;
  for (int i = 0; i < 1555; i++) {
    acc0 += siftData->h_data[i].coords2D[0] + siftData->h_data[i].data[7];
    acc1 += siftData1.h_data[i].coords2D[0] + siftData1.h_data[i].data[7];
  }
  if (std::fabs(acc0 - acc1) > 1e-6 * std::fabs(acc0)) ++failures;

  // ScaleDown + RootSIFT entry points
  cuImage half;
  half.Allocate(w / 2, h / 2, iAlignUp(w / 2, 128), true);
  ScaleDown(half, img1, 0.5f);
  half.DeviceToHost();
  if (!(half.h_data[0] >= 0.0f && half.h_data[0] <= 255.0f)) ++failures;
  siftData1.ConvertSiftToRootSift();
  siftData1.Synchronize();
  double l2 = 0;
  for (int k = 0; k < 128; k++) l2 += (double)siftData1.h_data[0].data[k] * siftData1.h_data[0].data[k];
  std::printf("RootSIFT |d|^2 of point 0: %.6f\n", l2);
  if (std::fabs(l2 - 1.0) > 1e-4) ++failures;
  {
    // ExtractRootSift (the reference's commented-out entry point, cuSIFT.cu:122-134): RootSIFT as the descriptor
    // kernel's epilogue must give the bits of ExtractSift + ConvertSiftToRootSift.  Compared on the deterministic
    // coarse-octave block (the octave-0 tail of the capped run is order-dependent).
    SiftData siftData2;
    InitSiftData(siftData2, 4096, true, true);
    ExtractRootSift(siftData2, img1, 6, 0.0f, 0.1f, 0.0f);
    int same = 0, matched = 0;
    for (int i = 0; i < 1555; i++) {
      const SiftPoint &a = siftData1.h_data[i];
      for (int j = 0; j < 1555; j++) {
        const SiftPoint &b = siftData2.h_data[j];
        if (a.coords2D[0] == b.coords2D[0] && a.coords2D[1] == b.coords2D[1] && a.scale == b.scale) {
          ++matched;
          same += std::memcmp(a.data, b.data, sizeof(a.data)) == 0;
          break;
        }
      }
    }
    std::printf("ExtractRootSift: %d points; coarse block %d matched, %d bit-identical descriptors\n", siftData2.numPts,
                matched, same);
    if (siftData2.numPts != siftData1.numPts || matched != 1555 || same != 1555) ++failures;
    FreeSiftData(siftData2);
  }
  FreeSiftData(siftData1);

  delete siftData;
  delete cuIm;
  cusift_dropin::shutdown();
  std::printf(failures ? "FAILED (%d)\n" : "PASSED\n", failures);
  return failures ? 1 : 0;
}
