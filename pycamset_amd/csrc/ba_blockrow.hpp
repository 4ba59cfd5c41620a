// ba_blockrow.hpp — products with a MATERIALISED block-row Jacobian that stays on the device (SURVEY 8 row f2 for generated chains).
//
// The three hand-fused chains have matrix-free products (ba_matfree.hpp: every 2 x P block is recomputed on the fly) and the
// block-reduced normal equations (ba_normal.hpp).  A generated chain — any composition, user blocks included — has one kernel that
// writes its dense block rows (ba_generic.hpp); what a Levenberg-Marquardt solver needs from them (optimisation_handling.py:88-98
// hands J to scipy: column norms, J^T f, lsmr mat-vecs) is generic over the chain once J exists: row i's column p belongs to
// block b(p) and sits at global column  start_b + np_b * index_b(detection) + (p - col0_b)  — the reference's
// get_block_param_inds (afb:192-233) evaluated on the fly from the detection's (camera, image, key).  One lane per detection,
// J read once per product (2 P doubles per detection: 432 B at P = 27), sums through a workgroup-private accumulator of the whole
// parameter string in LDS (ds_add_f64) while it fits 64 KB, one global f64 atomic per touched entry per workgroup after that; larger
// strings use global atomics directly.  J never crosses PCIe and, sharded, ranks all-reduce parameter-sized vectors only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ba_device.hpp"

namespace pcs {

constexpr int BLOCKROW_MAX_BLOCKS = 12;
constexpr int BR_JV = 0, BR_JTU = 1, BR_JTJV = 2, BR_DIAG = 3, BR_GRAD = 4;   // pcs_matfree's op codes

struct BlockRowArgs {
    DetTable tab;
    const double *J;       // 2N x P dense block rows, u row then v row
    const double *resid;   // N x 2 (BR_GRAD)
    const double *in;      // n_params (JV, JTJV) / 2N (JTU)
    double *out;           // 2N (JV) / n_params (zeroed by the host)
    double *cost;          // BR_GRAD: sum r^2 (zeroed by the host)
    int64_t n, n_params;
    int32_t P, n_blocks, lds_acc;
    int32_t blk_col0[BLOCKROW_MAX_BLOCKS], blk_np[BLOCKROW_MAX_BLOCKS], blk_link[BLOCKROW_MAX_BLOCKS];   // link: 0 camera, 1 image, 2 key
    int64_t blk_start[BLOCKROW_MAX_BLOCKS];
};

template <int OP>
__global__ __launch_bounds__(256) void blockrow_kernel(const BlockRowArgs a) {
    extern __shared__ double br_acc[];
    constexpr bool SCATTER = OP != BR_JV;
    if constexpr (SCATTER) {
        if (a.lds_acc) {
            for (int64_t q = threadIdx.x; q < a.n_params; q += blockDim.x) br_acc[q] = 0.0;
            __syncthreads();
        }
    }
    auto add = [&](const int64_t col, const double v) {
        if (a.lds_acc) atomicAdd(br_acc + col, v);
        else unsafeAtomicAdd(a.out + col, v);
    };
    double cost = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        int c, im, k;
        load_indices(a.tab, i, c, im, k);
        const double *ju = a.J + 2 * i * (int64_t)a.P, *jv = ju + a.P;
        double wu = 0.0, wv = 0.0;
        if constexpr (OP == BR_JV || OP == BR_JTJV) {
            for (int b = 0; b < a.n_blocks; ++b) {
                const int idx = a.blk_link[b] == 0 ? c : a.blk_link[b] == 1 ? im : k;
                const double *vin = a.in + a.blk_start[b] + (int64_t)a.blk_np[b] * idx;
                for (int q = 0; q < a.blk_np[b]; ++q) {
                    const int p = a.blk_col0[b] + q;
                    wu += ju[p] * vin[q];
                    wv += jv[p] * vin[q];
                }
            }
        }
        if constexpr (OP == BR_JV) {
            a.out[2 * i] = wu;
            a.out[2 * i + 1] = wv;
        }
        if constexpr (OP == BR_JTU) { wu = a.in[2 * i]; wv = a.in[2 * i + 1]; }
        if constexpr (OP == BR_GRAD) {
            wu = a.resid[2 * i];
            wv = a.resid[2 * i + 1];
            cost += wu * wu + wv * wv;
        }
        if constexpr (SCATTER) {
            for (int b = 0; b < a.n_blocks; ++b) {
                const int idx = a.blk_link[b] == 0 ? c : a.blk_link[b] == 1 ? im : k;
                const int64_t col = a.blk_start[b] + (int64_t)a.blk_np[b] * idx;
                for (int q = 0; q < a.blk_np[b]; ++q) {
                    const int p = a.blk_col0[b] + q;
                    const double v = OP == BR_DIAG ? ju[p] * ju[p] + jv[p] * jv[p] : ju[p] * wu + jv[p] * wv;
                    add(col + q, v);
                }
            }
        }
    }
    if constexpr (OP == BR_GRAD) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cost += __shfl_xor(cost, off);
        if ((threadIdx.x & 63) == 0 && cost != 0.0) unsafeAtomicAdd(a.cost, cost);
    }
    if constexpr (SCATTER) {
        if (a.lds_acc) {
            __syncthreads();
            for (int64_t q = threadIdx.x; q < a.n_params; q += blockDim.x) {
                const double v = br_acc[q];
                if (v != 0.0) unsafeAtomicAdd(a.out + q, v);
            }
        }
    }
}

}  // namespace pcs
