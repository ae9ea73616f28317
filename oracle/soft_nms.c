/*
 * ORACLE — test infrastructure only.  Never imported, linked or executed by the
 * product path (rrnet_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call into this file.
 *
 * CPU restatement of the reference's Soft-NMS:
 *   /root/reference/ext/nms/nms/cpu_nms.pyx:17-120  (cpu_soft_nms)
 * as called through /root/reference/ext/nms/nms_wrapper.py:13-19.
 *
 * The arithmetic follows the C that Cython generates from that .pyx, not its
 * surface text (checked against the generated code in the build container):
 *   - the literal `+ 1` is a double `+ 1.0`, so `area`, `iw`, `ih`, `ua` are
 *     float-float -> double +1.0 -> (double product / sum) -> rounded to float;
 *   - `iw * ih` is a float product, `ov = (iw*ih) / ua` a float division;
 *   - gaussian weight = (float) exp((double)(-(ov*ov) / sigma)), the quotient in float;
 *   - `weight * score` is a float product; comparisons are float compares;
 *   - no FMA contraction (x86-64 baseline): build with -ffp-contract=off.
 * Parity pin: tests/test_oracle_softnms.py checks this file against the README
 * known-answer vector (nms_wrapper.py:36-50) and against golden vectors produced
 * by the compiled reference (oracle/_ref) — tests/golden/softnms_*.npz.
 *
 * Rows are `stride` floats wide (>= 5); only columns 0..4 are permuted, exactly
 * as cpu_nms.pyx:55-66,109-113 do (column 5, the class, stays where it is).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

static float f_max(float a, float b) { return a >= b ? a : b; } /* cpu_nms.pyx:11-12 */
static float f_min(float a, float b) { return a <= b ? a : b; } /* cpu_nms.pyx:14-15 */

/* Returns N' (rows [0,N') are the kept detections, in selection order), or -1
 * where the reference raises ZeroDivisionError (ua == 0 or sigma == 0). */
int oracle_soft_nms(float *boxes, int n_in, int stride, float sigma, float Nt,
                    float threshold, unsigned int method)
{
    unsigned int N = (unsigned int)n_in;
    const unsigned int N0 = N; /* `for i in range(N)` is evaluated once, cpu_nms.pyx:36 */
#define B(r, c) boxes[(size_t)(r) * (size_t)stride + (c)]
    for (unsigned int i = 0; i < N0; ++i) {
        float maxscore = B(i, 4);
        int maxpos = (int)i;
        float tx1 = B(i, 0), ty1 = B(i, 1), tx2 = B(i, 2), ty2 = B(i, 3), ts = B(i, 4);
        int pos = (int)i + 1;
        while ((unsigned int)pos < N) {             /* first maximum wins: strict `<` (:49) */
            if (maxscore < B(pos, 4)) { maxscore = B(pos, 4); maxpos = pos; }
            pos++;
        }
        B(i, 0) = B(maxpos, 0); B(i, 1) = B(maxpos, 1); B(i, 2) = B(maxpos, 2);
        B(i, 3) = B(maxpos, 3); B(i, 4) = B(maxpos, 4);
        B(maxpos, 0) = tx1; B(maxpos, 1) = ty1; B(maxpos, 2) = tx2;
        B(maxpos, 3) = ty2; B(maxpos, 4) = ts;
        tx1 = B(i, 0); ty1 = B(i, 1); tx2 = B(i, 2); ty2 = B(i, 3); ts = B(i, 4);
        (void)ts;
        pos = (int)i + 1;
        while ((unsigned int)pos < N) {
            float x1 = B(pos, 0), y1 = B(pos, 1), x2 = B(pos, 2), y2 = B(pos, 3);
            float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
            float iw = (float)((double)(f_min(tx2, x2) - f_max(tx1, x1)) + 1.0);
            if (iw > 0.0f) {
                float ih = (float)((double)(f_min(ty2, y2) - f_max(ty1, y1)) + 1.0);
                if (ih > 0.0f) {
                    float ua = (float)(((((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0))
                                        + (double)area) - (double)(iw * ih));
                    if (ua == 0.0f) return -1;
                    float ov = (iw * ih) / ua;
                    float weight;
                    if (method == 1) {
                        weight = (ov > Nt) ? (float)(1.0 - (double)ov) : 1.0f;
                    } else if (method == 2) {
                        if (sigma == 0.0f) return -1;
                        float q = (-(ov * ov)) / sigma;
                        weight = (float)exp((double)q);
                    } else {
                        weight = (ov > Nt) ? 0.0f : 1.0f;
                    }
                    B(pos, 4) = weight * B(pos, 4);
                    if (B(pos, 4) < threshold) {     /* swap-with-last compaction (:108-115) */
                        B(pos, 0) = B(N - 1, 0); B(pos, 1) = B(N - 1, 1); B(pos, 2) = B(N - 1, 2);
                        B(pos, 3) = B(N - 1, 3); B(pos, 4) = B(N - 1, 4);
                        N = N - 1;
                        pos = pos - 1;
                    }
                }
            }
            pos = pos + 1;
        }
    }
#undef B
    return (int)N;
}

/* Batched form used by the tests / CPU baseline: `nseg` independent segments,
 * segment s holds rows [seg_off[s], seg_off[s+1]) of `boxes`; n_out[s] = N'. */
int oracle_soft_nms_segments(float *boxes, const int *seg_off, int nseg, int stride,
                             float sigma, float Nt, float threshold, unsigned int method,
                             int *n_out)
{
    for (int s = 0; s < nseg; ++s) {
        int n = seg_off[s + 1] - seg_off[s];
        int r = oracle_soft_nms(boxes + (size_t)seg_off[s] * stride, n, stride, sigma, Nt,
                                threshold, method);
        if (r < 0) return -1;
        n_out[s] = r;
    }
    return 0;
}

/* Hard NMS with the torchvision.ops.nms definition the reference calls at
 * /root/reference/models/rrnet.py:69,78 (torchvision is NOT vendored — parity
 * unpinned by the reference; torchvision-0.3 semantics: boxes visited in
 * descending score order, IoU without the +1 convention, suppress when
 * IoU > thresh).  `order` is the score-descending permutation (stable).
 * keep[] receives indices into the input; returns their count. */
int oracle_hard_nms(const float *boxes, int n, int stride, const int *order, float thresh,
                    int *keep)
{
    int nk = 0;
    unsigned char *supp = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    if (!supp) return -2;
    for (int a = 0; a < n; ++a) {
        int i = order[a];
        if (supp[i]) continue;
        keep[nk++] = i;
        const float *bi = boxes + (size_t)i * stride;
        float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
        for (int b = a + 1; b < n; ++b) {
            int j = order[b];
            if (supp[j]) continue;
            const float *bj = boxes + (size_t)j * stride;
            float xx1 = f_max(bi[0], bj[0]), yy1 = f_max(bi[1], bj[1]);
            float xx2 = f_min(bi[2], bj[2]), yy2 = f_min(bi[3], bj[3]);
            float w = f_max(0.0f, xx2 - xx1), h = f_max(0.0f, yy2 - yy1);
            float inter = w * h;
            float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
            float ovr = inter / (ai + aj - inter);
            if (ovr > thresh) supp[j] = 1;
        }
    }
    free(supp);
    return nk;
}
