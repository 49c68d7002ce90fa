// Internal interface of the strip kernel (cdl_strip.hip): the fused ISTA iteration for the shapes the shipped
// 2-D checkpoints use -- one image channel, stride 1 or 2, up to 192 subbands (CDLNet-s2030: K=30 M=169 P=7 s=2,
// /root/reference/trained_nets/CDLNet-s2030/args.json:2-9).  Reached through the cdl_fusedg_* entry points of
// include/cdlnet_hip.h (cdl_fusedg.hip dispatches here when cdl_strip_plan_for() accepts the geometry).
#pragma once
#include "cdl_common.h"

struct cdl_strip_plan {
    int P, S, MT, KS, KQ, RT;        // filter side, stride, 32-channel tiles, 16-tap / 16-channel k-steps, 32-slot tap tiles
    int Hz, Wz, nsx, nsy, SEG;       // code plane, 32-column strips, SEG-row segments
    int prows, pxw;                  // patch rows / columns per parity plane
    size_t items;                    // work items = N * nsy * nsx (one wave each)
    size_t frag_uint4;               // prepared weights of one (analysis-like, synthesis-like) pair
    size_t patch_floats, map_words;
};

bool cdl_strip_plan_for(const cdl_geom *g, cdl_strip_plan *pl);
// fragments of K pairs; shift2 as cdl_fusedg.hip's prep_pairs: 1 = (w1[k], w2[k+1]), 0 = (w1[k+1], w2[k])
int cdl_strip_prep_pairs(const cdl_geom *g, const cdl_strip_plan &pl, const float *const *w1, const float *const *w2,
                         int K, int shift2, void *frags, hipStream_t st);
// mode 0: z' = ST(zin + sgn * A r, tau), 1: the same without zin, 2: reverse stage (du = [map](zin + acc), dtau)
int cdl_strip_stage(const cdl_geom *g, const cdl_strip_plan &pl, int mode, const float *r, const float *zin,
                    const float *tau, const void *frags, float sgn, float *zout, float *patches, unsigned *map,
                    float *dtau_partial, int do_synth, int rev, int lay_in, int lay_out, hipStream_t st);
// floats of one code tensor in the row-strip channel-major layout (CDL_LAY_RSC): N * M * Hz * nsx * 32
size_t cdl_strip_rsc_floats(const cdl_geom *g, const cdl_strip_plan &pl);
int cdl_strip_assemble(const cdl_geom *g, const cdl_strip_plan &pl, const float *patches, const float *mask,
                       const float *sub, float alpha, float *out, hipStream_t st);

// ---- grouped shapes (cdl_stripg.hip): any C, 2-D / 3-D, unit stride, P in {3,5,7}, G = C * Pd in {1,3,5,7}, M <= 64 -- the
// shapes of cdl_fusedg.hip's tile kernel, whose prepared fragments (same MT, KS, KQ, G) this kernel reads
struct cdl_stripg_plan {
    int P, G, MT, KS, KQ;
    int nsx, nsy, SEG, prows, pxw;
    size_t items, patch_floats;
};
bool cdl_stripg_plan_for(const cdl_geom *g, cdl_stripg_plan *pl);
int cdl_stripg_stage(const cdl_geom *g, const cdl_stripg_plan &pl, int mode, const float *r, const float *zin,
                     const float *tau, const void *frags, float sgn, float *zout, float *patches, unsigned *map,
                     float *dtau_partial, int do_synth, int rev, int lay_in, int lay_out, hipStream_t st);
int cdl_stripg_assemble(const cdl_geom *g, const cdl_stripg_plan &pl, const float *patches, const float *mask,
                        const float *sub, float alpha, float *out, hipStream_t st);
size_t cdl_stripg_rsc_floats(const cdl_geom *g, const cdl_stripg_plan &pl);
