// fuse_rows.hip -- the REGISTER-ROWS kernel of the fused field query (gfx950, round 6): patch-resolution maps of 1024 fp32 channels
// (the reference's DINOv2 ViT-L features, fusion.py:694-697; BASELINE config 4).
//
// Every other family walks the POINTS and fetches four corner texels per (point, view) -- from the vector L1 (direct, cell runs:
// 64 B/clk per CU) or from an LDS pool (windows: 256 B/clk per CU, 128 KB of corner reads per point with 8 views x 1024 channels:
// 0.97 ms of LDS time alone on config 4's lattice).  On patch-resolution maps that traffic is almost all redundant: the 32 points of a
// small brick fall into two to four texel CELLS per view.  Here a workgroup keeps the fused ROWS of its 32 points in registers --
// lane l owns channels 4 l .. 4 l + 3 of every point: 32 x 4 = 128 accumulator VGPRs, v[128:255] -- and walks the CELLS: view by view
// (ascending, so every row still receives its views in the reference's order) and, inside a view, cell by cell, the four corner
// vectors of a cell are loaded ONCE (16 bytes per lane and corner, the next cell's underneath the current cell's arithmetic) and every
// point of the cell adds its four folded-weight fma to its own row.  Which row that is, is only known at run time; it is
// wave-uniform, so the accumulator is addressed through the VGPR INDEX MODE (s_set_gpr_idx_on: SRC2 and DST of v_pk_fma_f32 relative
// to M0 = 4 * point) -- no LDS, no moves.  Two points of a cell go through one block with the index switched between them (four
// independent fma chains instead of two).  Per (point, view) the operands and their order are those of gather_map's folded fast path
// (fuse_common.h): bit-identical to the other families.  Texel traffic per point: the cells' corners once (~16 KB) instead of 128 KB.
//
// Phase A is the window kernel's (one lane per (point, view), ordered view sums by DPP); it additionally sorts the valid pairs of
// every view by cell (rank by counting, 32 keys per view) into a flat list of OPS -- two points of a cell: {4 + 4 weights, the two row
// registers} -- and a list of cells {texel offset, ops}.  Points with a pair at the image border (a corner outside the map) and strict
// points (non-finite projection / maps not known finite) take rows_redo_point(): gather_map's arithmetic on global loads, after the
// loop.  Thin maps (the mask, colours) ride along through gather_map_u like in the window kernel.
// Index mode and the step blocks were measured on their own first: scripts/notebook/microbench/gpr_idx_fma.hip (bit-exact; 77 / 88
// TFLOP/s with one / two points per block at two waves per SIMD, 82 with static registers; fma_rate.hip: v_pk_fma_f32 itself reaches
// 118 / 135 TFLOP/s at two / four waves per SIMD).  Measured (MI355X, profiles/r6_sessions): config 4's lattice 1.68 -> 1.47 ms,
// its cloud 2.66 -> 2.18 ms, the 71 k surface points of the reference's shape 0.137 -> 0.125 ms; with FOUR views the window kernel
// stays ahead (2.07 vs 2.37 ms on the reference's lattice), so the planner sends only more than four views -- and what the windows do
// not take -- here (d3f_plan.h: rows_row).  A workgroup's 53 k cycles: KRt + phase A 8 k, ranks / ops 4 k, the cell loop 36 k (its
// v_pk_fma_f32 alone are 29 k of SIMD time when both resident waves are in it), row stores 4 k.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d3f_internal.h"
#include "d3f_device.h"
#include "fuse_common.h"

namespace d3f {

constexpr int kRowsPts = D3F_ROWS_PTS;                      // points per workgroup (x 4 accumulator registers per lane)
constexpr uint32_t kRowStrict = 1u;               // per-point state bits of phase A: strict point (reference order, full division)
constexpr uint32_t kRowBorder = 2u;               // a valid pair with a corner outside the map: the point goes through rows_redo_point
constexpr uint32_t kRowNoKey = 0xffffffffu;

// One OP of the loop: TWO points of a cell (a cell with an odd count ends with a point paired with itself at zero weights -- adding
// +-0 products to a row changes no bit) -- 48 bytes: two broadcast ds_read_b128 and one ds_read_b32 per op
struct __attribute__((aligned(16))) RowOp {
    float w[4];         // folded bilinear weights nw, ne, sw, se of the first point
    float x[4];         // ... of the second
    uint32_t ctrl;      // i1 | i2 << 8: 4 * slot of the point inside the brick = first accumulator register of its row
    uint32_t pad[3];
};

// The rows: v[128:255] -- point p's four channels are v[128 + 4 p .. 131 + 4 p].  The kernel is compiled with a budget of 128 VGPRs
// (amdgpu_num_vgpr), so the compiler never allocates these registers; only the asm blocks below name them (as clobbers, which is
// also what makes the kernel descriptor reserve 256).  (First form: the four 32-register tuples as "+{v[128:159]}" operands of every
// block -- correct, but with two kinds of blocks in a loop nest the allocator moved the tuples through scratch: 736 bytes of spills.)
// D3F_ROWS_PTS (d3f_internal.h) points per workgroup: 32 (rows in v[128:255], 128 VGPRs for the compiler, two waves per SIMD) or -- a build-time
// experiment -- 16 (rows in v[104:167], 104 VGPRs for the compiler, three waves per SIMD).  Measured (r6_s28, same box, 16 vs 32):
// config 4's lattice 1.511 vs 1.532 ms, its cloud 2.327 vs 2.281, the 71 k surface points 0.113 vs 0.123: a third wave per SIMD buys
// nothing where the kernel is bound by VALU issue (392 wave instructions per point, 256 of them v_pk_fma_f32) -- 32 stays.
#if D3F_ROWS_PTS == 32
#define D3F_ROWS_LO "v[128:129]"
#define D3F_ROWS_HI "v[130:131]"
#define D3F_ROWS_BUDGET 128
#define D3F_ROWS_WAVES 2
#define D3F_ROWS_CLOBBER \
    "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", \
    "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", \
    "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", \
    "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", \
    "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", \
    "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", \
    "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", \
    "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define D3F_ROWS_POINTS(X) \
    X(0, 128, 131) \
    X(1, 132, 135) \
    X(2, 136, 139) \
    X(3, 140, 143) \
    X(4, 144, 147) \
    X(5, 148, 151) \
    X(6, 152, 155) \
    X(7, 156, 159) \
    X(8, 160, 163) \
    X(9, 164, 167) \
    X(10, 168, 171) \
    X(11, 172, 175) \
    X(12, 176, 179) \
    X(13, 180, 183) \
    X(14, 184, 187) \
    X(15, 188, 191) \
    X(16, 192, 195) \
    X(17, 196, 199) \
    X(18, 200, 203) \
    X(19, 204, 207) \
    X(20, 208, 211) \
    X(21, 212, 215) \
    X(22, 216, 219) \
    X(23, 220, 223) \
    X(24, 224, 227) \
    X(25, 228, 231) \
    X(26, 232, 235) \
    X(27, 236, 239) \
    X(28, 240, 243) \
    X(29, 244, 247) \
    X(30, 248, 251) \
    X(31, 252, 255)
#define D3F_ROWS_ZERO_ASM \
    "v_mov_b64 v[128:129], 0\n\t" \
    "v_mov_b64 v[130:131], 0\n\t" \
    "v_mov_b64 v[132:133], 0\n\t" \
    "v_mov_b64 v[134:135], 0\n\t" \
    "v_mov_b64 v[136:137], 0\n\t" \
    "v_mov_b64 v[138:139], 0\n\t" \
    "v_mov_b64 v[140:141], 0\n\t" \
    "v_mov_b64 v[142:143], 0\n\t" \
    "v_mov_b64 v[144:145], 0\n\t" \
    "v_mov_b64 v[146:147], 0\n\t" \
    "v_mov_b64 v[148:149], 0\n\t" \
    "v_mov_b64 v[150:151], 0\n\t" \
    "v_mov_b64 v[152:153], 0\n\t" \
    "v_mov_b64 v[154:155], 0\n\t" \
    "v_mov_b64 v[156:157], 0\n\t" \
    "v_mov_b64 v[158:159], 0\n\t" \
    "v_mov_b64 v[160:161], 0\n\t" \
    "v_mov_b64 v[162:163], 0\n\t" \
    "v_mov_b64 v[164:165], 0\n\t" \
    "v_mov_b64 v[166:167], 0\n\t" \
    "v_mov_b64 v[168:169], 0\n\t" \
    "v_mov_b64 v[170:171], 0\n\t" \
    "v_mov_b64 v[172:173], 0\n\t" \
    "v_mov_b64 v[174:175], 0\n\t" \
    "v_mov_b64 v[176:177], 0\n\t" \
    "v_mov_b64 v[178:179], 0\n\t" \
    "v_mov_b64 v[180:181], 0\n\t" \
    "v_mov_b64 v[182:183], 0\n\t" \
    "v_mov_b64 v[184:185], 0\n\t" \
    "v_mov_b64 v[186:187], 0\n\t" \
    "v_mov_b64 v[188:189], 0\n\t" \
    "v_mov_b64 v[190:191], 0\n\t" \
    "v_mov_b64 v[192:193], 0\n\t" \
    "v_mov_b64 v[194:195], 0\n\t" \
    "v_mov_b64 v[196:197], 0\n\t" \
    "v_mov_b64 v[198:199], 0\n\t" \
    "v_mov_b64 v[200:201], 0\n\t" \
    "v_mov_b64 v[202:203], 0\n\t" \
    "v_mov_b64 v[204:205], 0\n\t" \
    "v_mov_b64 v[206:207], 0\n\t" \
    "v_mov_b64 v[208:209], 0\n\t" \
    "v_mov_b64 v[210:211], 0\n\t" \
    "v_mov_b64 v[212:213], 0\n\t" \
    "v_mov_b64 v[214:215], 0\n\t" \
    "v_mov_b64 v[216:217], 0\n\t" \
    "v_mov_b64 v[218:219], 0\n\t" \
    "v_mov_b64 v[220:221], 0\n\t" \
    "v_mov_b64 v[222:223], 0\n\t" \
    "v_mov_b64 v[224:225], 0\n\t" \
    "v_mov_b64 v[226:227], 0\n\t" \
    "v_mov_b64 v[228:229], 0\n\t" \
    "v_mov_b64 v[230:231], 0\n\t" \
    "v_mov_b64 v[232:233], 0\n\t" \
    "v_mov_b64 v[234:235], 0\n\t" \
    "v_mov_b64 v[236:237], 0\n\t" \
    "v_mov_b64 v[238:239], 0\n\t" \
    "v_mov_b64 v[240:241], 0\n\t" \
    "v_mov_b64 v[242:243], 0\n\t" \
    "v_mov_b64 v[244:245], 0\n\t" \
    "v_mov_b64 v[246:247], 0\n\t" \
    "v_mov_b64 v[248:249], 0\n\t" \
    "v_mov_b64 v[250:251], 0\n\t" \
    "v_mov_b64 v[252:253], 0\n\t" \
    "v_mov_b64 v[254:255], 0\n\t"
#else
#define D3F_ROWS_LO "v[104:105]"
#define D3F_ROWS_HI "v[106:107]"
#define D3F_ROWS_BUDGET 104
#define D3F_ROWS_WAVES 3
#define D3F_ROWS_CLOBBER \
    "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", \
    "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", \
    "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", \
    "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"
#define D3F_ROWS_POINTS(X) \
    X(0, 104, 107) \
    X(1, 108, 111) \
    X(2, 112, 115) \
    X(3, 116, 119) \
    X(4, 120, 123) \
    X(5, 124, 127) \
    X(6, 128, 131) \
    X(7, 132, 135) \
    X(8, 136, 139) \
    X(9, 140, 143) \
    X(10, 144, 147) \
    X(11, 148, 151) \
    X(12, 152, 155) \
    X(13, 156, 159) \
    X(14, 160, 163) \
    X(15, 164, 167)
#define D3F_ROWS_ZERO_ASM \
    "v_mov_b64 v[104:105], 0\n\t" \
    "v_mov_b64 v[106:107], 0\n\t" \
    "v_mov_b64 v[108:109], 0\n\t" \
    "v_mov_b64 v[110:111], 0\n\t" \
    "v_mov_b64 v[112:113], 0\n\t" \
    "v_mov_b64 v[114:115], 0\n\t" \
    "v_mov_b64 v[116:117], 0\n\t" \
    "v_mov_b64 v[118:119], 0\n\t" \
    "v_mov_b64 v[120:121], 0\n\t" \
    "v_mov_b64 v[122:123], 0\n\t" \
    "v_mov_b64 v[124:125], 0\n\t" \
    "v_mov_b64 v[126:127], 0\n\t" \
    "v_mov_b64 v[128:129], 0\n\t" \
    "v_mov_b64 v[130:131], 0\n\t" \
    "v_mov_b64 v[132:133], 0\n\t" \
    "v_mov_b64 v[134:135], 0\n\t" \
    "v_mov_b64 v[136:137], 0\n\t" \
    "v_mov_b64 v[138:139], 0\n\t" \
    "v_mov_b64 v[140:141], 0\n\t" \
    "v_mov_b64 v[142:143], 0\n\t" \
    "v_mov_b64 v[144:145], 0\n\t" \
    "v_mov_b64 v[146:147], 0\n\t" \
    "v_mov_b64 v[148:149], 0\n\t" \
    "v_mov_b64 v[150:151], 0\n\t" \
    "v_mov_b64 v[152:153], 0\n\t" \
    "v_mov_b64 v[154:155], 0\n\t" \
    "v_mov_b64 v[156:157], 0\n\t" \
    "v_mov_b64 v[158:159], 0\n\t" \
    "v_mov_b64 v[160:161], 0\n\t" \
    "v_mov_b64 v[162:163], 0\n\t" \
    "v_mov_b64 v[164:165], 0\n\t" \
    "v_mov_b64 v[166:167], 0\n\t"
#endif

__device__ __forceinline__ void rows_zero()
{
    asm volatile(D3F_ROWS_ZERO_ASM "" ::: D3F_ROWS_CLOBBER);
}

// row[p] += corners * weights with p wave-uniform: SRC2 and DST of every v_pk_fma_f32 are relative to M0 = 4 p (VGPR index mode);
// op_sel picks the weight inside its register pair.
// two points of one cell (or one point twice, the second time at zero weights): four chains, the index register switched between them
__device__ __forceinline__ void rows_step2(uint32_t i1, uint32_t i2, const f32x4 (&c)[4], f32x4 w, f32x4 x)
{
    const f32x2 a0 = {c[0].x, c[0].y}, a1 = {c[0].z, c[0].w}, b0 = {c[1].x, c[1].y}, b1 = {c[1].z, c[1].w};
    const f32x2 d0 = {c[2].x, c[2].y}, d1 = {c[2].z, c[2].w}, e0 = {c[3].x, c[3].y}, e1 = {c[3].z, c[3].w};
    const f32x2 w01 = {w.x, w.y}, w23 = {w.z, w.w}, x01 = {x.x, x.y}, x23 = {x.z, x.w};
    asm volatile("s_set_gpr_idx_on %[i1], 0xc\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_LO ", %[a0], %[w01], " D3F_ROWS_LO " op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_HI ", %[a1], %[w01], " D3F_ROWS_HI " op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_LO ", %[a0], %[x01], " D3F_ROWS_LO " op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_HI ", %[a1], %[x01], " D3F_ROWS_HI " op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i1]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_LO ", %[b0], %[w01], " D3F_ROWS_LO " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_HI ", %[b1], %[w01], " D3F_ROWS_HI " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_LO ", %[b0], %[x01], " D3F_ROWS_LO " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_HI ", %[b1], %[x01], " D3F_ROWS_HI " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i1]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_LO ", %[d0], %[w23], " D3F_ROWS_LO " op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_HI ", %[d1], %[w23], " D3F_ROWS_HI " op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_LO ", %[d0], %[x23], " D3F_ROWS_LO " op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_HI ", %[d1], %[x23], " D3F_ROWS_HI " op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i1]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_LO ", %[e0], %[w23], " D3F_ROWS_LO " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_HI ", %[e1], %[w23], " D3F_ROWS_HI " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_LO ", %[e0], %[x23], " D3F_ROWS_LO " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 " D3F_ROWS_HI ", %[e1], %[x23], " D3F_ROWS_HI " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_off"
                 :
                 : [i1] "s"(i1), [i2] "s"(i2), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [d0] "v"(d0), [d1] "v"(d1),
                   [e0] "v"(e0), [e1] "v"(e1), [w01] "v"(w01), [w23] "v"(w23), [x01] "v"(x01), [x23] "v"(x23)
                 : D3F_ROWS_CLOBBER);
}

// One point by gather_map's arithmetic (fuse_common.h), all 256 lanes on its 1024 channels: a strict point in the reference's order
// with the full division, any other (a pair at the image border) with the folded weights of the fast path.
__device__ __forceinline__ void rows_redo_point(const MapDesc &m, const EvalParams &P, const ViewRec *rec, float cnt, bool strict,
                                                uint32_t idx, uint32_t lane_off)
{
    using VT = f32x4;
    const int V = P.V;
    const char *__restrict__ data = reinterpret_cast<const char *>(m.data);
    const float denom = cnt + 1e-6f;                  // fusion.py:385
    VT acc = (VT)0.0f;
    for (int v = 0; v < V; ++v) {
        const ViewRec r = rec[v];
        if (!strict && r.valid == 0.0f) continue;      // exact skip (finite operands)
        const Corner c = corner_setup(m, r.gx, r.gy);
        const char *bv = data + (int64_t)v * m.sv * 4;
        const VT a = load_texel<4, false>(bv + (c.onw + lane_off)), b = load_texel<4, false>(bv + (c.one + lane_off));
        const VT d = load_texel<4, false>(bv + (c.osw + lane_off)), e = load_texel<4, false>(bv + (c.ose + lane_off));
        if (!strict) {
            float w0 = c.inw ? c.wnw : 0.0f, w1 = c.ine ? c.wne : 0.0f, w2 = c.isw ? c.wsw : 0.0f, w3 = c.ise ? c.wse : 0.0f;
            const float sc = fold_scale(r.wgt, cnt);
            w0 = w0 * sc; w1 = w1 * sc; w2 = w2 * sc; w3 = w3 * sc;
            acc = v_fma<VT>(a, w0, acc);
            acc = v_fma<VT>(b, w1, acc);
            acc = v_fma<VT>(d, w2, acc);
            acc = v_fma<VT>(e, w3, acc);
        } else {
            const VT av = c.inw ? a : (VT)0.0f, bvv = c.ine ? b : (VT)0.0f, dv = c.isw ? d : (VT)0.0f, ev = c.ise ? e : (VT)0.0f;
            VT s = av * c.wnw;
            s = v_fma<VT>(bvv, c.wne, s);
            s = v_fma<VT>(dv, c.wsw, s);
            s = v_fma<VT>(ev, c.wse, s);
            acc = acc + (s * r.valid) * r.wgt;         // fusion.py:385
        }
    }
    VT o = acc;                                        // folded weights carry 1/(cnt + 1e-6) already
    if (cnt == 0.0f) o = (VT)0.0f;                     // fusion.py:386
    else if (strict) o = strict_div<VT>(acc, denom);
    store_row_vec(reinterpret_cast<char *>(m.out) + ((uint64_t)idx * ((uint32_t)m.C * 4u) + lane_off), o);
}

__global__ __launch_bounds__(kBlock, D3F_ROWS_WAVES) __attribute__((amdgpu_num_vgpr(D3F_ROWS_BUDGET))) void fused_eval_rows_kernel(const EvalParams P)
{
    if (gated_out(P)) return;
    constexpr int TP = kRowsPts, NT = kBlock, MV = 8;
    using VT = f32x4;
    __shared__ float krt[MV * 12];
    __shared__ RowOp op_s[TP * MV + 1];          // (+1: the loop reads one op ahead)
    __shared__ uint32_t cell_off_s[TP * MV];      // byte offset of every cell's nw texel from the map's base (view included)
    __shared__ uint32_t cell_nops_s[TP * MV];     // ops of the cell
    __shared__ ViewRec rec_s[TP * MV];
    __shared__ __attribute__((aligned(16))) uint32_t key_s[MV][TP + 4];      // (+4: the views' rows on different banks, 16-byte reads)
    __shared__ uint32_t nvalid_s[MV], heads_s[MV], odd_s[MV];
    __shared__ float cnt_s[TP], aux_s[TP];
    __shared__ uint32_t flag_s[TP], idx_s[TP];
    __shared__ uint32_t redo_mask_s, dead_mask_s;

#ifdef D3F_EXPERIMENTS
    // phase stamps (D3F_EXP_STAMPS=1): lane 0 of wave 0 of every 64th workgroup writes s_memtime at the phase boundaries
    int stamp_k = 0;
    const bool stamping = P.exp_stamps != nullptr && (blockIdx.x & 63u) == 0u && threadIdx.x == 0 && (blockIdx.x >> 6) < 65536u;
#define D3F_STAMP() do { if (stamping && stamp_k < 31) P.exp_stamps[(size_t)(blockIdx.x >> 6) * 32 + 1 + stamp_k++] = __builtin_readcyclecounter(); } while (0)
#else
#define D3F_STAMP() do { } while (0)
#endif
    D3F_STAMP();                                    // 0: entry
    const int V = P.V;
    const MapDesc &m0 = P.maps[0];
    const bool walk = P.walk_nx > 0;
    // the brick of the lattice (bricks numbered z fastest, XCD k takes the k-th contiguous eighth), or TP consecutive points
    const int lbz = __ffs(P.walk_tz) - 1, lby = __ffs(P.walk_ty) - 1;
    int ox = 0, oy = 0, oz = 0;
    if (walk) {
        const uint32_t nbz = (uint32_t)((P.walk_nz + P.walk_tz - 1) / P.walk_tz), nby = (uint32_t)((P.walk_ny + P.walk_ty - 1) / P.walk_ty);
        const uint32_t b = (uint32_t)xcd_tile((int64_t)blockIdx.x, (int64_t)gridDim.x);
        const uint32_t bxy = b / nbz;
        oz = (int)(b - bxy * nbz) * P.walk_tz;
        const uint32_t bx = bxy / nby;
        oy = (int)(bxy - bx * nby) * P.walk_ty;
        ox = (int)bx * P.walk_tx;
    }
    const int bsx = min(P.walk_tx, P.walk_nx - ox), bsy = min(P.walk_ty, P.walk_ny - oy), bsz = min(P.walk_tz, P.walk_nz - oz);
    const int64_t tile_base = ((P.flags & kFlagXcdRemap) ? (int64_t)blockIdx.x : xcd_tile((int64_t)blockIdx.x, (int64_t)gridDim.x)) * TP;
    const int tile_n = walk ? TP : (int)min((int64_t)TP, P.n - tile_base);
    // slot p of the brick: its point (clipped slots repeat a neighbour) and whether the slot is a repeat
    auto slot_point = [&](int p, bool &dead, int &lx, int &ly, int &lz) -> int64_t {
        if (walk) {
            const int rz = p & (P.walk_tz - 1), ry = (p >> lbz) & (P.walk_ty - 1), rx = p >> (lbz + lby);
            dead = rz >= bsz || ry >= bsy || rx >= bsx;
            lz = min(rz, bsz - 1); ly = min(ry, bsy - 1); lx = min(rx, bsx - 1);
            return ((int64_t)(ox + lx) * P.walk_ny + (oy + ly)) * P.walk_nz + (oz + lz);
        }
        dead = p >= tile_n;
        lx = ly = lz = 0;
        const int64_t q = tile_base + min(p, tile_n - 1);
        return P.order ? min((int64_t)P.order[q], P.n - 1) : q;
    };
    const float mu = P.mu;
    const float Wm1 = (float)(P.W - 1), Hm1 = (float)(P.H - 1);

    // this lane's (point, view) of phase A; the point is requested NOW (a first touch of the point array: an HBM round trip under KRt)
    const int vp_log2 = V <= 1 ? 0 : (V <= 2 ? 1 : (V <= 4 ? 2 : 3));
    const int VP = 1 << vp_log2;
    const int pa_p = (int)threadIdx.x >> vp_log2, pa_v = (int)threadIdx.x & (VP - 1);
    const bool pa_lane = (int)threadIdx.x < TP * VP;      // (TP * VP <= 256: one pass)
    bool pa_dead = false;
    int64_t pa_i = 0;
    float pa_x = 0.0f, pa_y = 0.0f, pa_z = 0.0f;
    if (pa_lane) {
        int lx, ly, lz;
        pa_i = slot_point(pa_p, pa_dead, lx, ly, lz);
        if (pa_v < V) {
            if (walk && P.grid_x) { pa_x = P.grid_x[ox + lx]; pa_y = P.grid_y[oy + ly]; pa_z = P.grid_z[oz + lz]; }
            else fetch_point(P, pa_i, pa_x, pa_y, pa_z);
        }
    }
    rows_zero();                                    // (the rows' registers are nobody else's: zeroed under the loads)
    compute_krt(P.K, P.pose, V, krt, NT);
    if (threadIdx.x < MV) { nvalid_s[threadIdx.x] = 0u; heads_s[threadIdx.x] = 0u; odd_s[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) { redo_mask_s = 0u; dead_mask_s = 0u; }
    __syncthreads();
    D3F_STAMP();                                    // 1: KRt
    // ---- phase A: lane = (point, view), the views of a point adjacent ----
    const int lane = threadIdx.x & 63;
    const int base = lane & ~(VP - 1);
    const bool finite_maps = maps_are_finite(P);
    const uint32_t sxb = (uint32_t)m0.sx * 4u, syb = (uint32_t)m0.sy * 4u;
    uint32_t my_key = kRowNoKey, my_off = 0u;
    float my_w[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    {
        float dv = 0.0f, valid = 0.0f, wgt = 0.0f;
        uint32_t st = 0u;
        const bool dead = pa_dead;
        const int64_t i = pa_i;
        if (pa_lane) {
            if (pa_v < V) {
                const float px = pa_x, py = pa_y, pz = pa_z;
                const ViewOut o = eval_view<0>(P.depth, P.H, P.W, krt + pa_v * 12, pa_v, px, py, pz, Wm1, Hm1, mu, wgt);
                ViewRec r;
                r.gx = o.gx; r.gy = o.gy; r.wgt = wgt; r.valid = o.valid;
                rec_s[pa_p * V + pa_v] = r;
                dv = o.dist * o.valid;                                          // fusion.py:364 (product only)
                valid = o.valid;
                if (!(isfinite(o.gx) && isfinite(o.gy) && isfinite(wgt))) st |= kRowStrict;
                if (o.valid != 0.0f) {
                    // the corner set-up of corner_setup(), in texel coordinates
                    const float ix = unnormalize(o.gx, m0.fw), iy = unnormalize(o.gy, m0.fh);
                    const float x0 = floorf(ix), y0 = floorf(iy);
                    const float tx = ix - x0, ty = iy - y0;
                    const float ex = 1.0f - tx, sy = 1.0f - ty;
                    const bool inmap = x0 >= 0.0f && x0 <= (float)(m0.fw - 2) && y0 >= 0.0f && y0 <= (float)(m0.fh - 2);
                    if (inmap) {
                        const uint32_t cx = (uint32_t)(int)x0, cy = (uint32_t)(int)y0;
                        my_key = cy * (uint32_t)m0.fw + cx;
                        my_off = (uint32_t)((int64_t)pa_v * m0.sv * 4) + cy * syb + cx * sxb;
                        my_w[0] = sy * ex; my_w[1] = sy * tx; my_w[2] = ty * ex; my_w[3] = ty * tx;      // folded below
                    } else {
                        st |= kRowBorder;
                    }
                }
            }
        }
        // sums over the views in view order (fusion.py:364-370); every lane of the wave takes part
        float dsum, cnt;
        uint32_t stp;
        view_sums(V, base, dv, valid, st, dsum, cnt, stp);
        if (!finite_maps) stp |= kRowStrict;
        if (pa_lane && pa_v < V) {
            const bool fast = stp == 0u && !dead && my_key != kRowNoKey;
            if (fast) {
                const float sc = fold_scale(wgt, cnt);                           // folded weights (fuse_common.h)
                my_w[0] = my_w[0] * sc; my_w[1] = my_w[1] * sc; my_w[2] = my_w[2] * sc; my_w[3] = my_w[3] * sc;
            } else {
                my_key = kRowNoKey;
            }
            key_s[pa_v][pa_p] = my_key;
        }
        if (pa_lane && pa_v == 0) {
            const bool all_invalid = (cnt == 0.0f);                             // fusion.py:366
            float dist_out = dsum / (cnt + 1e-6f);
            if (all_invalid) dist_out = 1e3f;                                   // fusion.py:367
            cnt_s[pa_p] = cnt;
            idx_s[pa_p] = (uint32_t)i;
            flag_s[pa_p] = (stp & kRowStrict) ? 1u : 0u;
            aux_s[pa_p] = dist_out;
            if (dead) atomicOr(&dead_mask_s, 1u << pa_p);
            else if (stp != 0u) atomicOr(&redo_mask_s, 1u << pa_p);
        }
    }
    D3F_STAMP();                                    // 2: phase A (this wave)
    __syncthreads();
    D3F_STAMP();                                    // 3: ... every wave
    // ---- the valid pairs of every view sorted by cell: rank by counting over the view's 32 keys; a cell's points pair up into ops ----
    uint32_t my_rank = 0u, my_same = 0u, my_before = 0u;
    if (my_key != kRowNoKey) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q4 = 0; q4 < TP; q4 += 4) {
            const u32x4 kv = *reinterpret_cast<const u32x4 *>(&key_s[pa_v][q4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t kq = kv[j];
                const int q = q4 + j;
                my_rank += (kq < my_key || (kq == my_key && q < pa_p)) ? 1u : 0u;
                my_same += kq == my_key ? 1u : 0u;
                my_before += (kq == my_key && q < pa_p) ? 1u : 0u;
            }
        }
        atomicAdd(&nvalid_s[pa_v], 1u);
        if (my_before == 0u) {                         // the cell's first point: bit `rank` says a cell starts there / has an odd count
            atomicOr(&heads_s[pa_v], 1u << my_rank);
            if (my_same & 1u) atomicOr(&odd_s[pa_v], 1u << my_rank);
        }
    }
    __syncthreads();
    uint32_t ncells = 0u, nops = 0u;
    {
        uint32_t obase = 0u, cbase = 0u;
        for (int u = 0; u < V; ++u) {
            const uint32_t ou = (nvalid_s[u] + (uint32_t)__popc(odd_s[u])) >> 1, hu = (uint32_t)__popc(heads_s[u]);
            if (u < pa_v) { obase += ou; cbase += hu; }
            nops += ou; ncells += hu;
        }
        if (my_key != kRowNoKey) {
            const uint32_t h = my_rank - my_before;                              // rank of the cell's first point
            const uint32_t below = (1u << h) - 1u;
            RowOp *op = &op_s[obase + ((h + (uint32_t)__popc(odd_s[pa_v] & below)) >> 1) + (my_before >> 1)];
            const uint32_t p4 = 4u * (uint32_t)pa_p;
            if (my_before & 1u) {
                op->x[0] = my_w[0]; op->x[1] = my_w[1]; op->x[2] = my_w[2]; op->x[3] = my_w[3];
                reinterpret_cast<unsigned char *>(&op->ctrl)[1] = (unsigned char)p4;
            } else {
                op->w[0] = my_w[0]; op->w[1] = my_w[1]; op->w[2] = my_w[2]; op->w[3] = my_w[3];
                reinterpret_cast<unsigned char *>(&op->ctrl)[0] = (unsigned char)p4;
                if (my_before + 1u == my_same) {                                 // the last point of an odd cell: paired with itself at zero weights
                    op->x[0] = 0.0f; op->x[1] = 0.0f; op->x[2] = 0.0f; op->x[3] = 0.0f;
                    reinterpret_cast<unsigned char *>(&op->ctrl)[1] = (unsigned char)p4;
                }
            }
            if (my_before == 0u) {
                const uint32_t ci = cbase + (uint32_t)__popc(heads_s[pa_v] & below);
                cell_off_s[ci] = my_off; cell_nops_s[ci] = (my_same + 1u) >> 1;
            }
        }
    }
    __syncthreads();
    D3F_STAMP();                                    // 4: ranks, ops and cells listed
#ifdef D3F_EXPERIMENTS
    if (stamping) { P.exp_stamps[(size_t)(blockIdx.x >> 6) * 32 + 30] = ncells; P.exp_stamps[(size_t)(blockIdx.x >> 6) * 32 + 31] = nops; }
#endif

    // ---- the rows: cells in (view, cell) order.  FOUR register sets of corners, three cells requested ahead: a cell's loads are
    //      inline asm and waited for with a counted vmcnt (the compiler's own bookkeeping joins the loop's paths into vmcnt(0) in front
    //      of every block: each cell waited for the NEXT cell's texels), every set has exactly one defining asm, so nothing but the
    //      step blocks ever reads it.  All control is scalar and known early: lane l of two VGPRs holds cell l's texel offset and op
    //      count (v_readlane with a scalar index, no LDS round trip in the loop's control), the ops of a cell are a counted loop, and
    //      an op's 36 bytes are read one op ahead into one of two register sets.  (History: one cell ahead, 49 k of a brick's 67 k
    //      cycles in this loop; three ahead but cell ends found in the ops' own flags, 51 k -- a loop with NO arithmetic took 34 k.)
    const uint32_t lane_off = threadIdx.x * 16u;
    const char *__restrict__ data = reinterpret_cast<const char *>(m0.data);
    ncells = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncells);
    nops = (uint32_t)__builtin_amdgcn_readfirstlane((int)nops);
    if (nops > 0u) {
        struct OpRegs { f32x4 w, x; uint32_t c; };
        const uint32_t vo1 = lane_off + sxb, vo2 = lane_off + syb, vo3 = lane_off + syb + sxb;
        uint32_t bank_off = cell_off_s[lane], bank_nops = cell_nops_s[lane], bank_base = 0u;     // cells 0 .. 63 (more: reloaded)
        auto cell_bank = [&](uint32_t k) {
            if ((k & ~63u) != bank_base) { bank_base = k & ~63u; bank_off = cell_off_s[bank_base + lane]; bank_nops = cell_nops_s[bank_base + lane]; }
        };
        auto issue = [&](uint32_t k, VT (&c)[4]) {
            cell_bank(k);
            const char *b = data + (uint32_t)__builtin_amdgcn_readlane((int)bank_off, (int)(k & 63u));
            asm volatile("global_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %8\n\t"
                         "global_load_dwordx4 %2, %6, %8\n\tglobal_load_dwordx4 %3, %7, %8"
                         : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                         : "v"(lane_off), "v"(vo1), "v"(vo2), "v"(vo3), "s"(b)
                         : "memory");
        };
        // wait for a cell's four loads with `behind` younger cells (four loads each) still in flight
        auto wait_cell = [&](uint32_t behind) {
            if (behind >= 5u) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else if (behind == 4u) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (behind == 3u) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (behind == 2u) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (behind == 1u) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        auto read_op = [&](uint32_t t) -> OpRegs {
            const unsigned char *r = reinterpret_cast<const unsigned char *>(op_s) + t * (uint32_t)sizeof(RowOp);
            OpRegs o;
            o.w = *reinterpret_cast<const f32x4 *>(r); o.x = *reinterpret_cast<const f32x4 *>(r + 16);
            o.c = *reinterpret_cast<const uint32_t *>(r + 32);
            return o;
        };
        auto exec_op = [&](const OpRegs &o, const VT (&cur)[4]) {
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)o.c);
            rows_step2(c & 0xffu, (c >> 8) & 0xffu, cur, o.w, o.x);
        };
        uint32_t t = 0u, k = 0u;
        OpRegs op0 = read_op(0u), op1;
        // the n >= 1 ops of one cell on corner set `cur`: the next op is always in flight (op_s has one spare entry)
        auto run_cell = [&](const VT (&cur)[4], uint32_t n) {
            for (; n >= 2u; n -= 2u, t += 2u) {
                op1 = read_op(t + 1u);
                exec_op(op0, cur);
                op0 = read_op(t + 2u);
                exec_op(op1, cur);
            }
            if (n) {
                op1 = read_op(t + 1u);
                exec_op(op0, cur);
                op0 = op1;
                ++t;
            }
        };
        // k: the next cell to request; one part of the body requests cell k into one set and runs cell k - AHEAD from the next one
        // round the ring (D3F_ROWS_SETS = AHEAD + 1 register sets of 16 VGPRs).  Measured on config 4 (lattice / cloud, same box,
        // r6_s25): 4 sets 1.527 / 2.261 ms, 5 sets 1.553 / 2.298, 6 sets 1.493 / 2.181 -- box noise: three cells ahead is enough.
        // (Two ahead with the left neighbour's ne / se corners borrowed as nw / sw -- a quarter to a third fewer loads -- lost:
        // 1.78 / 2.79 ms, scripts/notebook/patches/r6_rows_left_neighbour.hip.txt.)
#ifndef D3F_ROWS_SETS
#define D3F_ROWS_SETS 4
#endif
        constexpr uint32_t AHEAD = D3F_ROWS_SETS - 1;
        VT c0[4], c1[4], c2[4], c3[4];
#if D3F_ROWS_SETS >= 5
        VT c4[4];
#endif
#if D3F_ROWS_SETS >= 6
        VT c5[4];
#endif
        const uint32_t last = ncells - 1u;
#define D3F_ROWS_PART(ISSUE_SET, RUN_SET)                                                                \
        {                                                                                                    \
            if (k < ncells) issue(k, ISSUE_SET);                                                             \
            if (k >= AHEAD) {                                                                                \
                const uint32_t kr = k - AHEAD;                                                               \
                cell_bank(kr);                                                                               \
                const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)bank_nops, (int)(kr & 63u));     \
                wait_cell(min(k, last) - kr);                                                                \
                run_cell(RUN_SET, n);                                                                        \
                if (kr == last) break;                                                                       \
            }                                                                                                \
            ++k;                                                                                             \
        }
        for (;;) {
#if D3F_ROWS_SETS == 4
            D3F_ROWS_PART(c0, c1) D3F_ROWS_PART(c1, c2) D3F_ROWS_PART(c2, c3) D3F_ROWS_PART(c3, c0)
#elif D3F_ROWS_SETS == 5
            D3F_ROWS_PART(c0, c1) D3F_ROWS_PART(c1, c2) D3F_ROWS_PART(c2, c3) D3F_ROWS_PART(c3, c4) D3F_ROWS_PART(c4, c0)
#else
            D3F_ROWS_PART(c0, c1) D3F_ROWS_PART(c1, c2) D3F_ROWS_PART(c2, c3) D3F_ROWS_PART(c3, c4) D3F_ROWS_PART(c4, c5) D3F_ROWS_PART(c5, c0)
#endif
        }
#undef D3F_ROWS_PART
    }
    D3F_STAMP();                                    // 5: the rows are summed
    // ---- rows out (non-temporal, write-through: store_row_vec), one uniform base per point ----
    const uint32_t skip = (uint32_t)__builtin_amdgcn_readfirstlane((int)(redo_mask_s | dead_mask_s));
    const uint32_t idx_bank = idx_s[lane & (TP - 1)];            // lane p: the global index of slot p (v_readlane below: no LDS round trip per row)
    const uint32_t row_bytes = (uint32_t)m0.C * 4u;
    char *const out_bytes = reinterpret_cast<char *>(m0.out);
#define D3F_ROWS_STORE(P_, LO, HI)                                                                                     \
    if (!((skip >> P_) & 1u)) {                                                                                        \
        const uint32_t ip = (uint32_t)__builtin_amdgcn_readlane((int)idx_bank, P_);                                    \
        const char *row = out_bytes + (uint64_t)ip * row_bytes;                                                        \
        asm volatile("global_store_dwordx4 %0, v[" #LO ":" #HI "], %1 sc1 nt" ::"v"(lane_off), "s"(row) : "memory");  \
    }
    D3F_ROWS_POINTS(D3F_ROWS_STORE)
#undef D3F_ROWS_STORE
    asm volatile("s_nop 1" ::: "memory");                // (the store-data wait states of store_row_vec, fuse_common.h)
    D3F_STAMP();                                    // 6: rows stored (issued)
    if ((int)threadIdx.x < TP) {                        // per-point outputs (clipped slots repeat a neighbour: same values twice)
        P.out_dist[idx_s[threadIdx.x]] = aux_s[threadIdx.x];
        P.out_valid[idx_s[threadIdx.x]] = cnt_s[threadIdx.x] == 0.0f ? 0 : 1;
    }
    // ---- points the loop left out: border pairs and strict points, gather_map's arithmetic ----
    uint32_t redo = (uint32_t)__builtin_amdgcn_readfirstlane((int)redo_mask_s);
    while (redo) {
        const int p = __ffs((int)redo) - 1;
        redo &= redo - 1u;
        rows_redo_point(m0, P, rec_s + p * V, cnt_s[p], flag_s[p] != 0u, idx_s[p], lane_off);
    }
    // ---- the other (thin) maps of the call ----
    for (int s = 1; s < P.n_maps; ++s) {
        const MapDesc &mt = P.maps[s];
        switch (mt.vw) {
        case 4: gather_map_u<4, false, true>(mt, P, rec_s, cnt_s, flag_s, idx_s, 0, TP, nullptr); break;
        case 2: gather_map_u<2, false, true>(mt, P, rec_s, cnt_s, flag_s, idx_s, 0, TP, nullptr); break;
        default: gather_map_u<1, false, true>(mt, P, rec_s, cnt_s, flag_s, idx_s, 0, TP, nullptr); break;
        }
    }
    D3F_STAMP();                                    // 7: redone points, thin maps
#ifdef D3F_EXPERIMENTS
    if (stamping) P.exp_stamps[(size_t)(blockIdx.x >> 6) * 32] = (unsigned long long)stamp_k;
#endif
#undef D3F_STAMP
}

hipError_t launch_rows(const EvalParams &P, hipStream_t stream)
{
    if (P.tile_pts != kRowsPts || P.maps[0].C != 1024 || P.maps[0].esize != 4 || P.V > 8) return hipErrorInvalidValue;
    int64_t ntiles = (P.n + P.tile_pts - 1) / P.tile_pts;
    if (P.walk_nx > 0) {
        if (P.walk_tx * P.walk_ty * P.walk_tz != kRowsPts) return hipErrorInvalidValue;
        ntiles = (int64_t)((P.walk_nx + P.walk_tx - 1) / P.walk_tx) * ((P.walk_ny + P.walk_ty - 1) / P.walk_ty) *
                 ((P.walk_nz + P.walk_tz - 1) / P.walk_tz);
    }
    hipLaunchKernelGGL(fused_eval_rows_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, stream, P);
    return hipGetLastError();
}

}  // namespace d3f
