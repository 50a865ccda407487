// gpr_idx_fma.hip -- does VGPR index mode (s_set_gpr_idx_on: SRC2_REL | DST_REL) work on v_pk_fma_f32 on gfx950, and what does a step
// cost?  A workgroup of 256 lanes keeps the rows of 32 points x 1024 channels in registers (lane l: channels 4l..4l+3 of every point,
// v[128:255]) and adds, per step, four corner vectors times four weights to the row of a point chosen AT RUN TIME (wave-uniform).
// Checks the result against the host (fmaf chains: bit-exact), then times a long list of steps on L1-resident corners.
//   hipcc --offload-arch=gfx950 -O3 gpr_idx_fma.hip -o gpr_idx_fma && ./gpr_idx_fma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

#define ACC_OPS "+{v[128:159]}"(A0), "+{v[160:191]}"(A1), "+{v[192:223]}"(A2), "+{v[224:255]}"(A3)

__device__ __forceinline__ void step(f32x32 &A0, f32x32 &A1, f32x32 &A2, f32x32 &A3, uint32_t idx4, f32x4 a, f32x4 b, f32x4 d, f32x4 e,
                                     f32x4 w)
{
    const f32x2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
    const f32x2 d0 = {d.x, d.y}, d1 = {d.z, d.w}, e0 = {e.x, e.y}, e1 = {e.z, e.w};
    const f32x2 w01 = {w.x, w.y}, w23 = {w.z, w.w};
    asm volatile("s_set_gpr_idx_on %[idx], 0xc\n\t"
                 "v_pk_fma_f32 v[128:129], %[a0], %[w01], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[a1], %[w01], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[b0], %[w01], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[b1], %[w01], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[d0], %[w23], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[d1], %[w23], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[e0], %[w23], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[e1], %[w23], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_off"
                 : ACC_OPS
                 : [idx] "s"(idx4), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [d0] "v"(d0), [d1] "v"(d1), [e0] "v"(e0),
                   [e1] "v"(e1), [w01] "v"(w01), [w23] "v"(w23));
}


// two points of the same cell in one block: four independent accumulator chains, the index register switched between them
__device__ __forceinline__ void step2(f32x32 &A0, f32x32 &A1, f32x32 &A2, f32x32 &A3, uint32_t i1, uint32_t i2, f32x4 a, f32x4 b, f32x4 d,
                                      f32x4 e, f32x4 w, f32x4 x)
{
    const f32x2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
    const f32x2 d0 = {d.x, d.y}, d1 = {d.z, d.w}, e0 = {e.x, e.y}, e1 = {e.z, e.w};
    const f32x2 w01 = {w.x, w.y}, w23 = {w.z, w.w}, x01 = {x.x, x.y}, x23 = {x.z, x.w};
    asm volatile("s_set_gpr_idx_on %[i1], 0xc\n\t"
                 "v_pk_fma_f32 v[128:129], %[a0], %[w01], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[a1], %[w01], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 v[128:129], %[a0], %[x01], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[a1], %[x01], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[b0], %[w01], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[b1], %[w01], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 v[128:129], %[b0], %[x01], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[b1], %[x01], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[d0], %[w23], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[d1], %[w23], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 v[128:129], %[d0], %[x23], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[d1], %[x23], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[e0], %[w23], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[e1], %[w23], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 v[128:129], %[e0], %[x23], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[e1], %[x23], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_off"
                 : ACC_OPS
                 : [i1] "s"(i1), [i2] "s"(i2), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [d0] "v"(d0), [d1] "v"(d1),
                   [e0] "v"(e0), [e1] "v"(e1), [w01] "v"(w01), [w23] "v"(w23), [x01] "v"(x01), [x23] "v"(x23));
}

// the same with the index switched after every FOUR instructions (a point's nw and ne terms together): half the switches,
// dependent instructions two apart
__device__ __forceinline__ void step2b(f32x32 &A0, f32x32 &A1, f32x32 &A2, f32x32 &A3, uint32_t i1, uint32_t i2, f32x4 a, f32x4 b, f32x4 d,
                                      f32x4 e, f32x4 w, f32x4 x)
{
    const f32x2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
    const f32x2 d0 = {d.x, d.y}, d1 = {d.z, d.w}, e0 = {e.x, e.y}, e1 = {e.z, e.w};
    const f32x2 w01 = {w.x, w.y}, w23 = {w.z, w.w}, x01 = {x.x, x.y}, x23 = {x.z, x.w};
    asm volatile("s_set_gpr_idx_on %[i1], 0xc\n\t"
                 "v_pk_fma_f32 v[128:129], %[a0], %[w01], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[a1], %[w01], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[b0], %[w01], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[b1], %[w01], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 v[128:129], %[a0], %[x01], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[a1], %[x01], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[b0], %[x01], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[b1], %[x01], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[d0], %[w23], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[d1], %[w23], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[e0], %[w23], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[e1], %[w23], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 v[128:129], %[d0], %[x23], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[d1], %[x23], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[e0], %[x23], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[e1], %[x23], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_off"
                 : ACC_OPS
                 : [i1] "s"(i1), [i2] "s"(i2), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [d0] "v"(d0), [d1] "v"(d1),
                   [e0] "v"(e0), [e1] "v"(e1), [w01] "v"(w01), [w23] "v"(w23), [x01] "v"(x01), [x23] "v"(x23));
}



// FOUR points of one cell in one block: eight independent chains
__device__ __forceinline__ void step4(f32x32 &A0, f32x32 &A1, f32x32 &A2, f32x32 &A3, uint32_t i1, uint32_t i2, uint32_t i3, uint32_t i4, f32x4 a, f32x4 b,
                                      f32x4 d, f32x4 e, f32x4 w, f32x4 x, f32x4 y, f32x4 z)
{
    const f32x2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
    const f32x2 d0 = {d.x, d.y}, d1 = {d.z, d.w}, e0 = {e.x, e.y}, e1 = {e.z, e.w};
    const f32x2 w01 = {w.x, w.y}, w23 = {w.z, w.w}, x01 = {x.x, x.y}, x23 = {x.z, x.w}, y01 = {y.x, y.y}, y23 = {y.z, y.w}, z01 = {z.x, z.y}, z23 = {z.z, z.w};
    asm volatile("s_set_gpr_idx_on %[i1], 0xc\n\t"
                 "v_pk_fma_f32 v[128:129], %[a0], %[w01], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[a1], %[w01], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 v[128:129], %[a0], %[x01], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[a1], %[x01], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i3]\n\t"
                 "v_pk_fma_f32 v[128:129], %[a0], %[y01], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[a1], %[y01], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i4]\n\t"
                 "v_pk_fma_f32 v[128:129], %[a0], %[z01], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[a1], %[z01], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[b0], %[w01], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[b1], %[w01], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 v[128:129], %[b0], %[x01], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[b1], %[x01], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i3]\n\t"
                 "v_pk_fma_f32 v[128:129], %[b0], %[y01], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[b1], %[y01], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i4]\n\t"
                 "v_pk_fma_f32 v[128:129], %[b0], %[z01], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[b1], %[z01], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[d0], %[w23], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[d1], %[w23], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 v[128:129], %[d0], %[x23], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[d1], %[x23], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i3]\n\t"
                 "v_pk_fma_f32 v[128:129], %[d0], %[y23], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[d1], %[y23], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i4]\n\t"
                 "v_pk_fma_f32 v[128:129], %[d0], %[z23], v[128:129] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[d1], %[z23], v[130:131] op_sel_hi:[1,0,1]\n\t"
                 "s_set_gpr_idx_idx %[i1]\n\t"
                 "v_pk_fma_f32 v[128:129], %[e0], %[w23], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[e1], %[w23], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i2]\n\t"
                 "v_pk_fma_f32 v[128:129], %[e0], %[x23], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[e1], %[x23], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i3]\n\t"
                 "v_pk_fma_f32 v[128:129], %[e0], %[y23], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[e1], %[y23], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_idx %[i4]\n\t"
                 "v_pk_fma_f32 v[128:129], %[e0], %[z23], v[128:129] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 v[130:131], %[e1], %[z23], v[130:131] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                 "s_set_gpr_idx_off"
                 : ACC_OPS
                 : [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3), [i4] "s"(i4), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [d0] "v"(d0), [d1] "v"(d1),
                   [e0] "v"(e0), [e1] "v"(e1), [w01] "v"(w01), [w23] "v"(w23), [x01] "v"(x01), [x23] "v"(x23), [y01] "v"(y01), [y23] "v"(y23), [z01] "v"(z01), [z23] "v"(z23));
}

// texels: [ntex][1024]; a step s reads the four texels tl[s][0..3] (one 'cell'), adds them with weights wl[s] to point pl[s]
__global__ __launch_bounds__(256, 2) void k(const float *__restrict__ tex, const int *__restrict__ pl, const int *__restrict__ tl,
                                            const f32x4 *__restrict__ wl, int nsteps, int reload_every, float *__restrict__ out)
{
    f32x32 A0 = (f32x32)0.0f, A1 = (f32x32)0.0f, A2 = (f32x32)0.0f, A3 = (f32x32)0.0f;
    const int l = threadIdx.x;
    const int *plb = pl + (size_t)blockIdx.x * nsteps, *tlb = tl + (size_t)blockIdx.x * nsteps * 4;
    const f32x4 *wlb = wl + (size_t)blockIdx.x * nsteps;
    f32x4 a = (f32x4)0.0f, b = a, d = a, e = a;
    for (int s = 0; s < nsteps; ++s) {
        if (s % reload_every == 0) {          // uniform: a new cell
            const int t0 = __builtin_amdgcn_readfirstlane(tlb[4 * s]), t1 = __builtin_amdgcn_readfirstlane(tlb[4 * s + 1]);
            const int t2 = __builtin_amdgcn_readfirstlane(tlb[4 * s + 2]), t3 = __builtin_amdgcn_readfirstlane(tlb[4 * s + 3]);
            a = reinterpret_cast<const f32x4 *>(tex + (size_t)t0 * 1024)[l];
            b = reinterpret_cast<const f32x4 *>(tex + (size_t)t1 * 1024)[l];
            d = reinterpret_cast<const f32x4 *>(tex + (size_t)t2 * 1024)[l];
            e = reinterpret_cast<const f32x4 *>(tex + (size_t)t3 * 1024)[l];
        }
        const int p = __builtin_amdgcn_readfirstlane(plb[s]);
        if (reload_every > 2 && s + 1 < nsteps && (s + 1) % reload_every != 0) {      // pairs of steps on one cell: two index values per block
            const int p2 = __builtin_amdgcn_readfirstlane(plb[s + 1]);
            if (p2 != p) {
                step2(A0, A1, A2, A3, (uint32_t)p * 4u, (uint32_t)p2 * 4u, a, b, d, e, wlb[s], wlb[s + 1]);
                ++s;
                continue;
            }
        }
        step(A0, A1, A2, A3, (uint32_t)p * 4u, a, b, d, e, wlb[s]);
    }
    float *ob = out + (size_t)blockIdx.x * 32 * 1024;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        reinterpret_cast<f32x4 *>(ob + (size_t)(p)*1024)[l] = f32x4{A0[4 * p], A0[4 * p + 1], A0[4 * p + 2], A0[4 * p + 3]};
        reinterpret_cast<f32x4 *>(ob + (size_t)(p + 8) * 1024)[l] = f32x4{A1[4 * p], A1[4 * p + 1], A1[4 * p + 2], A1[4 * p + 3]};
        reinterpret_cast<f32x4 *>(ob + (size_t)(p + 16) * 1024)[l] = f32x4{A2[4 * p], A2[4 * p + 1], A2[4 * p + 2], A2[4 * p + 3]};
        reinterpret_cast<f32x4 *>(ob + (size_t)(p + 24) * 1024)[l] = f32x4{A3[4 * p], A3[4 * p + 1], A3[4 * p + 2], A3[4 * p + 3]};
    }
}


// the same loop with nothing but the steps: point index from arithmetic, weights and corners loop-invariant registers
template <int MODE>     // 0: index mode on/off per step; 1: no index mode (always point 0): the plain v_pk_fma_f32 rate
__global__ __launch_bounds__(256, 2) void k_pure(const float *__restrict__ tex, int nsteps, float *__restrict__ out)
{
    f32x32 A0 = (f32x32)0.0f, A1 = (f32x32)0.0f, A2 = (f32x32)0.0f, A3 = (f32x32)0.0f;
    const int l = threadIdx.x;
    const f32x4 a = reinterpret_cast<const f32x4 *>(tex)[l], b = reinterpret_cast<const f32x4 *>(tex + 1024)[l];
    const f32x4 d = reinterpret_cast<const f32x4 *>(tex + 2048)[l], e = reinterpret_cast<const f32x4 *>(tex + 3072)[l];
    const f32x4 w = {0.25f, 0.5f, 0.125f, 0.0625f};
    uint32_t p4 = 0u;
    for (int s = 0; s < nsteps; ++s) {
        p4 = (p4 + 28u) & 127u;
        asm volatile("" : "+s"(p4));
        if (MODE == 2) { step2(A0, A1, A2, A3, p4, p4 ^ 64u, a, b, d, e, w, w); ++s; }
        else if (MODE == 4) { step4(A0, A1, A2, A3, p4, p4 ^ 64u, p4 ^ 32u, p4 ^ 96u, a, b, d, e, w, w, w, w); s += 3; }
        else if (MODE == 3) { step2b(A0, A1, A2, A3, p4, p4 ^ 64u, a, b, d, e, w, w); ++s; }
        else step(A0, A1, A2, A3, MODE == 0 ? p4 : 0u, a, b, d, e, w);
    }
    float *ob = out + (size_t)blockIdx.x * 32 * 1024;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        reinterpret_cast<f32x4 *>(ob + (size_t)(p)*1024)[l] = f32x4{A0[4 * p], A0[4 * p + 1], A0[4 * p + 2], A0[4 * p + 3]};
        reinterpret_cast<f32x4 *>(ob + (size_t)(p + 8) * 1024)[l] = f32x4{A1[4 * p], A1[4 * p + 1], A1[4 * p + 2], A1[4 * p + 3]};
        reinterpret_cast<f32x4 *>(ob + (size_t)(p + 16) * 1024)[l] = f32x4{A2[4 * p], A2[4 * p + 1], A2[4 * p + 2], A2[4 * p + 3]};
        reinterpret_cast<f32x4 *>(ob + (size_t)(p + 24) * 1024)[l] = f32x4{A3[4 * p], A3[4 * p + 1], A3[4 * p + 2], A3[4 * p + 3]};
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main()
{
    const int ntex = 96, nb_check = 8;
    std::vector<float> tex((size_t)ntex * 1024);
    srand(7);
    for (auto &x : tex) x = (float)rand() / RAND_MAX - 0.5f;
    auto run = [&](int nblocks, int nsteps, int reload_every, bool check) -> int {
        std::vector<int> pl((size_t)nblocks * nsteps), tl((size_t)nblocks * nsteps * 4);
        std::vector<float> wl((size_t)nblocks * nsteps * 4);
        for (size_t s = 0; s < pl.size(); ++s) {
            pl[s] = rand() % 32;
            for (int c = 0; c < 4; ++c) { tl[4 * s + c] = rand() % ntex; wl[4 * s + c] = (float)rand() / RAND_MAX; }
        }
        float *d_tex, *d_out; int *d_pl, *d_tl; f32x4 *d_wl;
        CK(hipMalloc(&d_tex, tex.size() * 4)); CK(hipMalloc(&d_out, (size_t)nblocks * 32 * 1024 * 4));
        CK(hipMalloc(&d_pl, pl.size() * 4)); CK(hipMalloc(&d_tl, tl.size() * 4)); CK(hipMalloc(&d_wl, wl.size() * 4));
        CK(hipMemcpy(d_tex, tex.data(), tex.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_pl, pl.data(), pl.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_tl, tl.data(), tl.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_wl, wl.data(), wl.size() * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), 0, 0, d_tex, d_pl, d_tl, d_wl, nsteps, reload_every, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (!check && rep == 2) {
                const double steps = (double)nblocks * nsteps;
                printf("blocks %6d steps %6d reload every %3d: %.3f ms  %.1f cycles(2.4GHz)/step/SIMD at 2 waves per SIMD x 256 CUs, %.1f TFLOP/s\n",
                       nblocks, nsteps, reload_every, ms, ms * 1e-3 * 2.4e9 / (steps / (256.0 * 2.0)) , steps * 4.0 * 1024.0 * 2.0 / (ms * 1e-3) / 1e12);
            }
        }
        if (check) {
            std::vector<float> out((size_t)nblocks * 32 * 1024);
            CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (int blk = 0; blk < nblocks; ++blk) {
                std::vector<float> ref((size_t)32 * 1024, 0.0f);
                int cell[4] = {0, 0, 0, 0};
                for (int s = 0; s < nsteps; ++s) {
                    const size_t g = (size_t)blk * nsteps + s;
                    if (s % reload_every == 0) for (int c = 0; c < 4; ++c) cell[c] = tl[4 * g + c];
                    float *r = &ref[(size_t)pl[g] * 1024];
                    for (int ch = 0; ch < 1024; ++ch)
                        for (int c = 0; c < 4; ++c) r[ch] = fmaf(tex[(size_t)cell[c] * 1024 + ch], wl[4 * g + c], r[ch]);
                }
                for (size_t i = 0; i < ref.size(); ++i)
                    if (ref[i] != out[(size_t)blk * 32 * 1024 + i]) ++bad;
            }
            printf("check: %d blocks x %d steps (reload every %d): %zu mismatching values of %zu\n", nblocks, nsteps, reload_every, bad, out.size());
            if (bad) return 2;
        }
        hipFree(d_tex); hipFree(d_out); hipFree(d_pl); hipFree(d_tl); hipFree(d_wl);
        return 0;
    };
    if (int r = run(nb_check, 200, 1, true)) return r;
    if (int r = run(nb_check, 256, 8, true)) return r;
    if (int r = run(nb_check, 256, 4, true)) return r;

    {
        float *d_tex, *d_out;
        const int nblocks = 4096, nsteps = 4096;
        CK(hipMalloc(&d_tex, tex.size() * 4)); CK(hipMalloc(&d_out, (size_t)nblocks * 32 * 1024 * 4));
        CK(hipMemcpy(d_tex, tex.data(), tex.size() * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int mode = 0; mode < 5; ++mode)
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_pure<0>, dim3(nblocks), dim3(256), 0, 0, d_tex, nsteps, d_out);
                else if (mode == 4) hipLaunchKernelGGL(k_pure<4>, dim3(nblocks), dim3(256), 0, 0, d_tex, nsteps, d_out);
                else if (mode == 3) hipLaunchKernelGGL(k_pure<3>, dim3(nblocks), dim3(256), 0, 0, d_tex, nsteps, d_out);
                else if (mode == 2) hipLaunchKernelGGL(k_pure<2>, dim3(nblocks), dim3(256), 0, 0, d_tex, nsteps, d_out);
                else hipLaunchKernelGGL(k_pure<1>, dim3(nblocks), dim3(256), 0, 0, d_tex, nsteps, d_out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double steps = (double)nblocks * nsteps;
                if (rep == 2) printf("pure loop, %s: %.3f ms, %.1f cycles(2.4GHz) per wave step of 8 v_pk_fma_f32 (SIMD time, two waves interleaved), %.1f TFLOP/s\n",
                                     mode == 0 ? "index mode per step" : (mode == 2 ? "two points per block, index switched every 2" : (mode == 4 ? "FOUR points per block, index switched every 2" : mode == 3 ? "two points per block, index switched every 4" : "static registers")), ms, ms * 1e-3 * 2.4e9 / (steps * 4.0 / 1024.0), steps * 8192.0 / (ms * 1e-3) / 1e12);
            }
    }
    for (int re : {1, 2, 4, 8, 16}) if (int r = run(512 * 8, 2048, re, false)) return r;
    return 0;
}
