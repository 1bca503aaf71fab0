/*
 * oracle/nms_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never shipped, never measured
 * except as bench.py's `cpu_baseline` leg).
 *
 * Plain-C restatement of the two third-party leaves the reference's NMS front-ends call:
 *   - torchvision.ops.nms            (reference call sites scripts/utils/metrics.py:385,421 and
 *                                     scripts/utils/nms.py:66,102)
 *   - dense pairwise IoU             (reference scripts/utils/metrics.py:138-164 `box_iou`)
 *
 * torchvision==0.10.1 (environment.yml:28) is NOT vendored in /root/reference and is not installed in
 * this image, so the greedy kernel below restates the PUBLISHED algorithm of torchvision 0.10.1's CPU
 * kernel (torchvision/csrc/ops/cpu/nms_kernel.cpp, `nms_kernel_impl<float>`):
 *     order = argsort(scores, descending); areas = (x2-x1)*(y2-y1);
 *     for i in order: if suppressed: continue; keep i;
 *        for later j: w = max(0, min(x2)-max(x1)); h = ...; inter = w*h;
 *                     ovr = inter / (area_i + area_j - inter); if (ovr > iou_threshold) suppress j
 * with `ovr` float and `iou_threshold` DOUBLE (the C++ signature takes double), i.e. the comparison is
 * carried out in double.  Tie order of equal scores is implementation-defined upstream; this oracle (and
 * the HIP path) define it as STABLE: equal scores keep ascending input order.
 *
 * "parity unpinned" for this leaf: the reference's tests never assert NMS output;
 * the wrapper logic around it IS pinned by tests/golden (generated through the reference's own Python).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no FMA contraction so float results are the
 * plain IEEE single-precision sequence the torch CPU path produces).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static void merge_sort_desc(const float *key, int64_t *idx, int64_t *tmp, int64_t n)
{
    /* bottom-up stable merge sort of idx by key descending */
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n;
            int64_t hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                /* take from the right run only when strictly greater => stable */
                if (key[idx[b]] > key[idx[a]]) tmp[o++] = idx[b++];
                else tmp[o++] = idx[a++];
            }
            while (a < mid) tmp[o++] = idx[a++];
            while (b < hi) tmp[o++] = idx[b++];
        }
        memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
    }
}

/* stable descending argsort; out[n] */
void oracle_argsort_desc(const float *key, int64_t n, int64_t *out)
{
    int64_t *tmp = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) out[i] = i;
    merge_sort_desc(key, out, tmp, n);
    free(tmp);
}

/* Greedy NMS.  boxes: n x 4 (x1,y1,x2,y2) row-major float; returns number kept, indices (into the
 * input order) written to keep[] in descending-score order.
 * cls may be NULL; when given, only pairs with cls[i]==cls[j] interact (== per-class NMS, the
 * `_batched_nms_vanilla` strategy of torchvision.ops.boxes.batched_nms). */
int64_t oracle_nms(const float *boxes, const float *scores, const float *cls, int64_t n,
                   double iou_threshold, int64_t *keep)
{
    if (n <= 0) return 0;
    int64_t *order = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    float *areas = (float *)malloc((size_t)n * sizeof(float));
    unsigned char *sup = (unsigned char *)calloc((size_t)n, 1);
    oracle_argsort_desc(scores, n, order);
    for (int64_t i = 0; i < n; ++i) {
        const float *b = boxes + 4 * i;
        areas[i] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    int64_t nk = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = order[_i];
        if (sup[i]) continue;
        keep[nk++] = i;
        const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2],
                    iy2 = boxes[4 * i + 3], ia = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j];
            if (sup[j]) continue;
            if (cls && cls[i] != cls[j]) continue;
            const float *bj = boxes + 4 * j;
            float xx1 = ix1 > bj[0] ? ix1 : bj[0];
            float yy1 = iy1 > bj[1] ? iy1 : bj[1];
            float xx2 = ix2 < bj[2] ? ix2 : bj[2];
            float yy2 = iy2 < bj[3] ? iy2 : bj[3];
            float w = xx2 - xx1; if (!(w > 0.0f)) w = 0.0f;
            float h = yy2 - yy1; if (!(h > 0.0f)) h = 0.0f;
            float inter = w * h;
            float ovr = inter / (ia + areas[j] - inter);
            if ((double)ovr > iou_threshold) sup[j] = 1;
        }
    }
    free(order); free(areas); free(sup);
    return nk;
}

/* Dense IoU, reference scripts/utils/metrics.py:138-164:
 *   inter = prod(clamp(min(rb) - max(lt), 0)); iou = inter / (area1[:,None] + area2 - inter) */
void oracle_box_iou(const float *a, int64_t n, const float *b, int64_t m, float *out)
{
    for (int64_t i = 0; i < n; ++i) {
        const float *p = a + 4 * i;
        float a1 = (p[2] - p[0]) * (p[3] - p[1]);
        for (int64_t j = 0; j < m; ++j) {
            const float *q = b + 4 * j;
            float a2 = (q[2] - q[0]) * (q[3] - q[1]);
            float rx = (p[2] < q[2] ? p[2] : q[2]) - (p[0] > q[0] ? p[0] : q[0]);
            float ry = (p[3] < q[3] ? p[3] : q[3]) - (p[1] > q[1] ? p[1] : q[1]);
            if (!(rx > 0.0f)) rx = 0.0f;   /* clamp(0); NaN stays NaN in torch, irrelevant here */
            if (!(ry > 0.0f)) ry = 0.0f;
            float inter = rx * ry;
            out[i * m + j] = inter / (a1 + a2 - inter);
        }
    }
}
