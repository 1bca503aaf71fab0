// Patch-staged weight gradient of the 3x3 convs (wgrad3.hip), shared with the planner / launcher in conv.hip.
#pragma once
#include "gfx950_dma.h"

// one unit of work of a grouped weight-gradient launch: (job, dw tile, pixel split); job == ~0u: padding of an XCD queue
struct WItem { unsigned job, tile, zz, pad; };

// One job of k_wgrad3 = the weight gradient of one 3x3 / pad 1 / stride 1 or 2 conv (fp16 operands, fp32 partial sums).
// The output rows of all images form ONE list of "virtual rows" u = n * UP + oh with UP = OH + 2 (stride 1) or OH + 1 (stride 2):
// the rows oh >= OH of an image do not exist (their dy is fetched as zeros), which makes the padded input rows an output row
// needs simply V = s * u + {0, 1, 2} with V = n * XP + (ih + 1), XP = s * UP -- no image-boundary case anywhere in the kernel.
struct W3P {
    const void* x; const void* dy;
    int B, XH, XW, ldx, C;         // x: input activations (NHWC, channel stride ldx)
    int OH, OW, ldy, N;            // dy: output gradient, N = Cout
    int s, K;                      // stride; K = 9 * C (row length of dw)
    int UP, XP;                    // virtual rows per image: output rows / padded input rows
    unsigned NU;                   // B * UP
    // step geometry: a step = RPS output rows x TC columns of a column strip (PX = RPS * TC pixels in nsub 16-pixel sub-steps)
    int TC, RPS, PX, nsub, strips;
    // workgroup tile of dw: NB x CB blocks of (32 output channels) x (32 input channels x 9 taps); SL = 8 / (NB * CB) of the
    // EIGHT wavefronts share a block and split its sub-steps (summed through LDS at the end)
    int NB, CB, NP, SL;
    int tn, tc;                    // tiles along N / along C
    // x window of a step: nrows padded input rows, each ppr DMA pieces (1 KiB) = [c-block][column planes][pixel][32 channels]
    int nrows, ppr, rowpitch;
    int plo, ple;                  // bytes of a c-block's column planes: stride 1: plo = (TC + 2) * 64, ple = 0;
                                   // stride 2: odd input columns (TC + 1 pixels) then even input columns (TC pixels)
    int xstage, stage;             // bytes of the x window / of one LDS stage (x window + NB dy planes of nsub * 1024 bytes)
    unsigned x_bytes, y_bytes;     // buffer descriptor extents (< 1 GiB each: the loader adds a row base and a lane offset
                                   // that may each be "out of range" on their own)
    unsigned uch, uranges;         // virtual rows per item (a multiple of RPS; an item walks them once per column strip), items per tile
    unsigned long long ws_off;     // split-K workspace of this layer, in floats
    unsigned zz0;                  // first partial slot of this job
    FastDiv dXP, dUP, dTC, dPPR;
    double step_cost;              // modelled cycles of one step (host side: item balancing)
};

/* 0 = `d` is a conv k_wgrad3 runs (fills the geometry of `p`, not its split); != 0: use the generic kernel.  any_route: the
 * geometry whatever the routing switches say (introspection) */
int w3_fill(const ayolo_conv_desc* d, const void* x, const void* dy, W3P& p, bool any_route = false);
/* cut the job into items of about `steps` steps each (all strips of a row range together; at least one item); p.uch / p.uranges */
void w3_split(W3P& p, double steps);
static inline unsigned w3_splits(const W3P& p) { return p.uranges; }
static inline unsigned w3_tiles(const W3P& p) { return (unsigned)(p.tn * p.tc); }
/* floats of one partial slot: tile-major, [tile][n-block][c-block][tap][32 rows][32 channels] (every block whole: edge tiles padded) */
static inline unsigned long long w3_slot_floats(const W3P& p) { return (unsigned long long)p.tn * p.tc * p.NB * p.CB * 9ull * 1024ull; }
static inline unsigned long long w3_item_steps(const W3P& p, unsigned ur) {
    const unsigned u0 = ur * p.uch, u1 = u0 + p.uch < p.NU ? u0 + p.uch : p.NU;
    return (unsigned long long)p.strips * ((u1 - u0 + (unsigned)p.RPS - 1) / (unsigned)p.RPS);
}
size_t w3_lds_bytes(const W3P& p);
/* grouped launch (jobs / items in device memory, pv ignored) or single job by value (items == nullptr: blocks = tiles * splits) */
int w3_launch(const W3P& pv, const W3P* jobs, const WItem* items, unsigned blocks, size_t lds, float* ws, int rp, hipStream_t s);
/* the row-pitch instantiation a job can run on: its window row pitch when the stride is 1 and an instantiation exists, else 0 */
static inline int w3_rp_class(const W3P& p) {
    return (p.s == 1 && (p.rowpitch == 2048 || p.rowpitch == 3072 || p.rowpitch == 4096 || p.rowpitch == 6144)) ? p.rowpitch : 0;
}
