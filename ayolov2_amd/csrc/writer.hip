// Result rows for the COCO json of a validation run, on the device: for every detection of a batch
// the un-letterboxing of scripts/utils/general.py:324-358 `scale_coords` (as ResultWriterTorch.scale_coords calls it,
// scripts/utils/multi_queue.py:316-339: gain and padding recomputed from the original image shape), the clip to the
// original image, xyxy -> [x, y, width, height] (multi_queue.py:262-266) and the class -> COCO category id table
// (multi_queue.py:78-159).  One launch and one device->host copy per batch replace the reference's per-image numpy work in
// a consumer process; the arithmetic is the reference's float32 sequence (this file is built with -ffp-contract=off).
#include "common.h"

__global__ __launch_bounds__(256) void k_coco_rows(const float* det, const int* img, long long n, const float* lb, const int* cat,
                                                   int ncat, float* out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float* d = det + i * 6;
        const float* L = lb + (long long)img[i] * 6;        // gain, pad_w, pad_h, w0, h0, scaled?
        float x1 = d[0], y1 = d[1], x2 = d[2], y2 = d[3];
        if (L[5] != 0.0f) {
            x1 = (x1 - L[1]) / L[0]; x2 = (x2 - L[1]) / L[0];
            y1 = (y1 - L[2]) / L[0]; y2 = (y2 - L[2]) / L[0];
            x1 = fminf(fmaxf(x1, 0.0f), L[3]); x2 = fminf(fmaxf(x2, 0.0f), L[3]);
            y1 = fminf(fmaxf(y1, 0.0f), L[4]); y2 = fminf(fmaxf(y2, 0.0f), L[4]);
            x2 = x2 - x1;                                    // width, height
            y2 = y2 - y1;
        }
        const int c = (int)d[5];
        float* o = out + i * 6;
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = d[4];
        o[5] = (float)((cat && c >= 0 && c < ncat) ? cat[c] : c);
    }
}

extern "C" int ayolo_coco_rows(const float* det, const int* img_of_row, int64_t n, const float* letterbox, const int* cat_table,
                               int ncat, float* out, ayolo_stream s) {
    if (n == 0) return AYOLO_OK;
    AY_CHECK_ARG(det && img_of_row && letterbox && out && n > 0, "coco_rows: bad args");
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_coco_rows, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, det, img_of_row, (long long)n, letterbox,
                       cat_table, ncat, out);
    AY_CHECK_LAUNCH("k_coco_rows");
    return AYOLO_OK;
}
