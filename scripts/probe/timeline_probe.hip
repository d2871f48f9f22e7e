// scripts/probe/timeline_probe.hip — where does a latency-bound kernel spend its microseconds?
//
// Builds the PRODUCT kernels (fc_small.hip, conv3x3_img.h) with -DSABER_TIMELINE, which makes wave 0 of every workgroup
// store the 100 MHz wall clock at a few phase boundaries, launches them cold (after a kernel that streams 192 MB, so
// the XCD L2s hold none of the operands; the 256 MB Infinity Cache may) and warm (back to back), and prints per phase
// the min / median / max over workgroups relative to the first stamp of the launch, next to the hipEvent time.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSABER_TIMELINE -I anakin_amd/csrc \
//         scripts/probe/timeline_probe.hip -o scripts/probe/timeline_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__device__ unsigned long long* saber_tl_buf = nullptr;

#include "../../anakin_amd/csrc/fc_small.hip"
#include "../../anakin_amd/csrc/elementwise.hip"
#include "../../anakin_amd/csrc/img_e1.hip"
#include "../../anakin_amd/csrc/igemm_m0_e1.hip"
#include "../../anakin_amd/csrc/igemm_m0_e2.hip"
#include "../../anakin_amd/csrc/igemm_m3_e3.hip"
#include "../../anakin_amd/csrc/igemm_m2_e3.hip"
#include "../../anakin_amd/csrc/igemm_dma_m0_e1.hip"
#include "../../anakin_amd/csrc/halo_e1.hip"
#include "../../anakin_amd/csrc/stem_pool.hip"
#include "../../anakin_amd/csrc/conv1x1_chain.hip"
#include "../../anakin_amd/csrc/conv_chain_coop.hip"
#include "../../anakin_amd/csrc/conv_stage_coop.hip"
namespace saber_mi355x {
void tile_dims(int tile, int* bm_k, int* bn_pix) {
    static const int d[TILE_COUNT][2] = {{32, 32}, {64, 32}, {64, 64}, {128, 64}, {64, 128}, {128, 128}};
    *bm_k = d[tile][0];
    *bn_pix = d[tile][1];
}
}  // namespace saber_mi355x

using namespace saber_mi355x;
static void launch_softmax_rows(int rows, int cols, const float* x, float* y, hipStream_t s) { (void)launch_softmax_f32(rows, cols, x, y, s); }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void flush_kernel(const uint4* big, size_t n, unsigned* out) {
    unsigned s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = big[i]; s ^= v.x ^ v.w; }
    if (s == 0x12345u) out[0] = s;
}

__global__ __launch_bounds__(256) void null_kernel(unsigned* out) {
    if (out == nullptr) out[threadIdx.x] = 0;
}

static void* dalloc(size_t bytes, int fill) {
    void* p;
    CK(hipMalloc(&p, bytes));
    std::vector<unsigned char> h(bytes);
    for (size_t i = 0; i < bytes; ++i) h[i] = fill < 0 ? (unsigned char)(rand() & 0xff) : (unsigned char)fill;
    CK(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice));
    return p;
}

struct Probe {
    hipStream_t st;
    hipEvent_t e0, e1;
    uint4* big;
    size_t nbig;
    unsigned* out;
    unsigned long long* tl;
    int maxblocks = 8192;
};

template <typename F>
static void run(Probe& P, const char* name, int blocks, int nph, F launch) {
    std::vector<unsigned long long> h((size_t)blocks * 16);
    for (int cold = 1; cold >= 0; --cold) {
        std::vector<float> ev;
        std::vector<std::vector<double>> ph(nph);   // per phase: all (block, rep) samples, us since the launch's first stamp
        for (int rep = 0; rep < 12; ++rep) {
            if (cold) hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, P.st, P.big, P.nbig, P.out);
            else launch();
            CK(hipMemsetAsync(P.tl, 0, (size_t)blocks * 128, P.st));
            CK(hipEventRecord(P.e0, P.st));
            launch();
            CK(hipEventRecord(P.e1, P.st));
            CK(hipEventSynchronize(P.e1));
            float ms;
            CK(hipEventElapsedTime(&ms, P.e0, P.e1));
            if (rep < 2) continue;
            ev.push_back(ms * 1000.f);
            CK(hipMemcpy(h.data(), P.tl, (size_t)blocks * 128, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (int b = 0; b < blocks; ++b) if (h[(size_t)b * 16]) t0 = std::min(t0, h[(size_t)b * 16]);
            for (int b = 0; b < blocks; ++b)
                for (int i = 0; i < nph; ++i)
                    if (h[(size_t)b * 16 + i]) ph[i].push_back((double)(h[(size_t)b * 16 + i] - t0) * 0.01);
        }
        std::sort(ev.begin(), ev.end());
        float chain = 0;
        {   // 40 dependent launches between one event pair: the steady-state cost per launch of a same-stream chain
            launch();
            CK(hipEventRecord(P.e0, P.st));
            for (int i = 0; i < 40; ++i) launch();
            CK(hipEventRecord(P.e1, P.st));
            CK(hipEventSynchronize(P.e1));
            CK(hipEventElapsedTime(&chain, P.e0, P.e1));
            chain *= 1000.f / 40;
        }
        printf("%-44s %s  hipEvent median %.2f us (min %.2f); chain of 40: %.2f us per launch\n", name, cold ? "cold" : "warm",
               ev[ev.size() / 2], ev[0], chain);
        for (int i = 0; i < nph; ++i) {
            if (ph[i].empty()) continue;
            std::sort(ph[i].begin(), ph[i].end());
            printf("    phase %2d: min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us   (%zu samples)\n", i, ph[i].front(),
                   ph[i][ph[i].size() / 2], ph[i][ph[i].size() * 9 / 10], ph[i].back(), ph[i].size());
        }
    }
}

int main(int argc, char** argv) {
    Probe P;
    CK(hipStreamCreate(&P.st));
    CK(hipEventCreate(&P.e0));
    CK(hipEventCreate(&P.e1));
    P.nbig = (size_t)192 << 16;   // 192 MB of uint4
    P.big = (uint4*)dalloc(P.nbig * 16, 3);
    P.out = (unsigned*)dalloc(4096, 0);
    CK(hipMalloc((void**)&P.tl, (size_t)P.maxblocks * 128));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(saber_tl_buf), &P.tl, sizeof(P.tl)));
    void* zero = dalloc(256, 0);

    run(P, "empty kernel (256 WGs)", 256, 1, [&] { hipLaunchKernelGGL(null_kernel, dim3(256), dim3(256), 0, P.st, P.out); });
    if (argc > 1 && !strcmp(argv[1], "coop")) {
        // 3x3-led chain at C = 256, 14 x 14: one workgroup per tile with 8 waves (tile code 3) against two cooperating workgroups
        // (conv_chain_coop.hip). Coop phases: 0 entry, 1 DMA + ring landed, 2 3x3 done + tile stored, 3 past the first pair barrier,
        // 4 first 1x1 done + y1 stored, 5 past the second barrier, 6 done
        for (int n : {8, 1}) {
            const int c = 256, hw = 14, K1 = 1024, M = n * hw * hw;
            ChainKArgs a;
            memset(&a, 0, sizeof a);
            a.M = M; a.in_u8 = 1; a.relu1 = 0; a.res_relu = 1; a.coeff_conv = 16.f; a.scale_conv = 0.05f; a.coeff_res = 16.f;
            a.scale_res = 0.04f; a.relu2 = 1; a.out_u8_2 = 1;
            a.x = dalloc((size_t)M * c, -1); a.res = dalloc((size_t)M * K1, -1);
            a.prm1 = dalloc((size_t)K1 * 12, 0); a.prm2 = dalloc((size_t)c * 12 + 2048, 0);
            a.y1 = dalloc((size_t)M * K1, 0); a.y2 = dalloc((size_t)M * c, 0);
            a.wstream = dalloc((size_t)2 * K1 * c + (size_t)9 * c * c + 65536, -1);
            auto magic = [](int d) { return d >= 2 ? (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d) : 0u; };
            a.prm0 = dalloc((size_t)c * 12 + 2048, 0); a.zero = zero; a.N = n; a.H = a.W = hw;
            a.tiles_x = 1; a.tiles_per_img = hw; a.mg_tiles_x = magic(1); a.mg_tpi = magic(hw); a.in0_u8 = 1; a.relu0 = 1;
            a.s0 = 1;
            const int tiles = n * hw;
            char nm[96];
            snprintf(nm, sizeof nm, "conv3x3+chain C=256 14x14 b%d 1x16 w8", n);
            run(P, nm, tiles, 6, [&] { launch_conv1x1_chain(a, c, K1, c, 3, 1, P.st); });
            CoopKArgs ck;
            ck.c = a;
            ck.coop_ctr = (unsigned long long*)dalloc((size_t)tiles * 32 * 8, 0);
            ck.coop_xcc = (unsigned*)dalloc((size_t)tiles * 32 * 4, 0);
            ck.coop_xch = dalloc((size_t)tiles * 16 * c, 0);
            ck.coop_err = (unsigned*)dalloc(64, 0);
            ck.n_tiles = tiles;
            snprintf(nm, sizeof nm, "conv3x3+chain C=256 14x14 b%d coop2", n);
            run(P, nm, (tiles + 7) / 8 * 16, 7, [&] { launch_conv_chain_coop(ck, P.st); });
            // conv_stage_coop.hip: four cooperating workgroups per tile of 2 rows x 16 columns; one block with the tiles spread over the
            // XCDs (the chain's tile code 15), one block with an image per XCD, five blocks in one launch. Stamps (of the LAST block
            // where a launch has several): 0 entry, 1 first halo + constants + 20 fragments landed, 2 3x3 done + arrival 1 + barrier,
            // 3 past wait 1, 4 1x1 + eltwise done, y1 stored, arrival 2, 5 past wait 2, 6 y2 stored, 7 past the last image barrier +
            // the next halo landed
            const int tiles4 = n * ((hw + 1) / 2);
            std::vector<StageBlk> hb(5);
            for (auto& B : hb) {
                memset(&B, 0, sizeof B);
                B.wstream = a.wstream; B.prm0 = a.prm0; B.prm1 = a.prm1; B.prm2 = a.prm2;
                B.coeff_conv = 16.f; B.scale_conv = 0.05f; B.coeff_res = 16.f; B.scale_res = 0.04f;
                B.in0_u8 = 1; B.relu0 = 1; B.in_u8 = 1; B.relu1 = 0; B.res_relu = 1; B.relu2 = 1; B.out_u8_2 = 1;
            }
            StageBlk* dblk;
            CK(hipMalloc((void**)&dblk, sizeof(StageBlk) * 5));
            CK(hipMemcpy(dblk, hb.data(), sizeof(StageBlk) * 5, hipMemcpyHostToDevice));
            Stage4KArgs<STAGE4_SHORT> sk;
            memset(&sk, 0, sizeof sk);
            sk.x = a.x; sk.res = a.res; sk.zero = zero; sk.blk = dblk;
            sk.grp_ctr = (unsigned long long*)dalloc((size_t)tiles4 * 32 * 8, 0);
            sk.img_ctr = (unsigned long long*)dalloc((size_t)n * ((hw + 1) / 2 + 1) * 16 * 8, 0);
            sk.xch = dalloc((size_t)tiles4 * 32 * c, 0);
            sk.xcc = (unsigned*)dalloc((size_t)tiles4 * 32 * 4, 0);
            sk.err = ck.coop_err;
            sk.N = n; sk.H = sk.W = hw; sk.tiles_x = 1; sk.tiles_per_img = (hw + 1) / 2;
            sk.mg_tiles_x = magic(1); sk.mg_tpi = magic(sk.tiles_per_img); sk.mg_wpi = magic(sk.tiles_per_img * 4);
            for (int i = 0; i < 5; ++i) {      // ping-pong outputs: block i + 1 reads block i's y2 (and keeps its y1 tile in LDS)
                sk.y1[i] = i % 2 ? a.y1 : dalloc((size_t)M * K1, 0);
                sk.y2[i] = i % 2 ? a.y2 : dalloc((size_t)M * c, 0);
            }
            sk.nblk = 1; sk.per_image = 0;
            snprintf(nm, sizeof nm, "stage4 C=256 14x14 b%d 1 block, tiles over XCDs", n);
            run(P, nm, (tiles4 + 7) / 8 * 32, 8, [&] { launch_conv_stage4(sk, P.st); });
            sk.per_image = 1;
            snprintf(nm, sizeof nm, "stage4 C=256 14x14 b%d 1 block, image per XCD", n);
            run(P, nm, (n + 7) / 8 * sk.tiles_per_img * 32, 8, [&] { launch_conv_stage4(sk, P.st); });
            sk.nblk = 5;
            snprintf(nm, sizeof nm, "stage4 C=256 14x14 b%d 5 blocks, image per XCD", n);
            run(P, nm, (n + 7) / 8 * sk.tiles_per_img * 32, 8, [&] { launch_conv_stage4(sk, P.st); });
            unsigned errs = 0;
            CK(hipMemcpy(&errs, ck.coop_err, 4, hipMemcpyDeviceToHost));
            printf("    coop error word: %u\n", errs);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "f32pw")) {
        // FP32 pointwise layers of ResNet50's res2 / res3 at batch 8 (2.4 TB/s where streaming reaches 4 - 6.5): the implicit-GEMM kernel on the
        // f32 MFMA (MODE 2) and on the bf16 planes (MODE 3), with and without the in-place sum. phases: 0 entry, 1 gather state set up, 2 first stage in LDS,
        // 5 reduction loop done, 4 stored
        struct P2 { const char* name; int mode, HW, C, K, tile, ks, sum; };
        const P2 ps[] = {
            {"f32 mfma   res2 64->256 @56 + sum in place  32x32 k2", 2, 56, 64, 256, TILE_32x32, 2, 1},
            {"f32 mfma   res2 64->256 @56                 32x32 k2", 2, 56, 64, 256, TILE_32x32, 2, 0},
            {"f32 mfma   res2 64->256 @56 + sum in place  64x64 k2", 2, 56, 64, 256, TILE_64x64, 2, 1},
            {"f32 planes res2 64->256 @56 + sum in place  64x64 k1", 3, 56, 64, 256, TILE_64x64, 1, 1},
            {"f32 mfma   res2 256->64 @56                 32x32 k2", 2, 56, 256, 64, TILE_32x32, 2, 0},
            {"f32 planes res3 128->512 @28 + sum in place 64x64 k1", 3, 28, 128, 512, TILE_64x64, 1, 1},
        };
        for (const P2& g : ps) {
            ConvKArgs a;
            memset(&a, 0, sizeof a);
            const int N = 8, kg = g.C, kgp = (kg + 63) / 64 * 64, kpad = (g.K + 127) / 128 * 128;
            a.N = N; a.H = a.W = a.OH = a.OW = g.HW; a.C = g.C; a.K = g.K; a.kh = a.kw = 1;
            a.stride_h = a.stride_w = a.dil_h = a.dil_w = 1;
            a.M = N * g.HW * g.HW; a.Kg = kg; a.Kg_pad = kgp; a.epi = EPI_F32; a.out_dtype = DT_F32; a.relu = 1;
            a.inv_ohw = 1.f / (g.HW * g.HW); a.inv_ow = 1.f / g.HW;
            a.x = dalloc((size_t)a.M * g.C * 4, 0);
            a.w = dalloc((size_t)(g.mode == 3 ? 3 * 2 : 4) * kpad * kgp + 65536, 0);
            a.w_plane_chunks = (int)((size_t)kpad * kgp / 8);
            a.zero = zero;
            a.y = dalloc((size_t)a.M * g.K * 4, 0);
            a.bias = (const float*)dalloc(2048 * 4, 0);
            a.res_mode = g.sum ? RES_SUM_INPLACE : RES_NONE;
            int bmk, bnp;
            tile_dims(g.tile, &bmk, &bnp);
            const int blocks = ((a.M + bnp - 1) / bnp) * ((g.K + bmk - 1) / bmk);
            const int estage = (g.mode == 3 ? 32 : 16) * g.ks;
            a.steps = (kg + estage - 1) / estage;
            if (g.mode == 3) run(P, g.name, blocks, 6, [&] { launch_igemm_m3_e3(g.tile, g.ks, a, P.st); });
            else run(P, g.name, blocks, 6, [&] { launch_igemm_m2_e3(g.tile, g.ks, a, P.st); });
        }
        return 0;
    }
    const bool stem_only = argc > 1 && !strcmp(argv[1], "stem");
    if (argc > 1 && !strcmp(argv[1], "chain")) {
        // conv1x1 chain: phases 0 entry, 1 loads / DMA issued, 2 first group's MFMAs done + DMA barrier, 3 first conv done,
        // 4 tile stored / second conv's operand in registers, 5 done
        struct { const char* name; int c, hw, n, tn, w3; } cs[] = {
            {"conv3x3+chain C=64 56x56 b8 2x16", 64, 56, 8, 2, 1}, {"conv3x3+chain C=128 28x28 b8 1x16", 128, 28, 8, 1, 1},
            {"conv3x3+conv1x1 C=64 56x56 b8 2x16", 64, 56, 8, 2, 2}, {"chain C=256 14x14 b8 px16 split2", 256, 14, 8, 9, 0},
            {"conv3x3+chain C=64 56x56 b1 2x16", 64, 56, 1, 2, 1},
            {"chain C=64 56x56 b8 px64", 64, 56, 8, 4}, {"chain C=64 56x56 b8 px32", 64, 56, 8, 2},
            {"chain C=128 28x28 b8 px32", 128, 28, 8, 2}, {"chain C=128 28x28 b8 px16", 128, 28, 8, 1},
            {"chain C=256 14x14 b8 px16", 256, 14, 8, 1}, {"chain C=512 7x7 b8 px16", 512, 7, 8, 1},
            {"chain C=64 56x56 b1 px32", 64, 56, 1, 2}, {"chain C=256 14x14 b1 px16", 256, 14, 1, 1}};
        for (auto& g : cs) {
            ChainKArgs a;
            memset(&a, 0, sizeof a);
            const int M = g.n * g.hw * g.hw, K1 = 4 * g.c;
            a.M = M; a.in_u8 = 1; a.relu1 = 0; a.res_relu = 1; a.coeff_conv = 16.f; a.scale_conv = 0.05f; a.coeff_res = 16.f;
            a.scale_res = 0.04f; a.relu2 = 1; a.out_u8_2 = 1;
            a.x = dalloc((size_t)M * g.c, -1); a.res = dalloc((size_t)M * K1, -1);
            a.wstream = dalloc((size_t)2 * K1 * g.c, -1);
            a.prm1 = dalloc((size_t)K1 * 12, 0); a.prm2 = dalloc((size_t)g.c * 12 + 1024, 0);
            a.y1 = dalloc((size_t)M * K1, 0); a.y2 = dalloc((size_t)M * g.c, 0);
            a.wstream = dalloc((size_t)4 * K1 * g.c + (size_t)9 * g.c * g.c + 65536, -1);
            const int tn = g.tn & 7;
            int blocks = ((M + 16 * tn - 1) / (16 * tn)) * ((g.tn & 8) ? 2 : 1);
            if (g.w3) {
                auto magic = [](int d) { return d >= 2 ? (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d) : 0u; };
                a.prm0 = dalloc((size_t)g.c * 12 + 1024, 0); a.zero = zero; a.N = g.n; a.H = a.W = g.hw;
                a.tiles_x = (g.hw + 15) / 16; a.tiles_per_img = a.tiles_x * ((g.hw + tn - 1) / tn);
                a.mg_tiles_x = magic(a.tiles_x); a.mg_tpi = magic(a.tiles_per_img); a.in0_u8 = 1; a.relu0 = 1;
                blocks = a.tiles_per_img * g.n;
            }
            run(P, g.name, blocks, 6, [&] { launch_conv1x1_chain(a, g.c, K1, g.w3 == 2 ? 0 : g.c, g.tn, g.w3 ? 1 : 0, P.st); });
        }
        return 0;
    }
    if (!stem_only) {   // ---- fc1000 of ResNet50, batch 8: phases 0 entry, 1 loads issued, 2 MFMAs done, 3 after the reduce barrier, 4 stored
        ConvKArgs a;
        memset(&a, 0, sizeof a);
        a.M = 8; a.C = 2048; a.K = 1000; a.Kg_pad = 2048; a.N = 8; a.H = a.W = a.OH = a.OW = 1; a.kh = a.kw = 1;
        a.x = dalloc(8 * 2048, -1); a.w = dalloc(1024 * 2048, -1); a.zero = zero;
        a.y = dalloc(8 * 1000 * 4, 0); a.scale = (const float*)dalloc(1024 * 4, 0); a.bias = (const float*)dalloc(1024 * 4, 0);
        a.epi = EPI_I8_FC_S8;
        run(P, "fc_i8_small 8x2048 -> 1000 (63 WGs)", 63, 6, [&] { launch_fc_i8_small(a, P.st); });
        // round 5: the same fc with the Softmax over its output in the launch (phases 5 = this workgroup arrived LAST and starts to
        // normalise, 6 = probabilities stored: one sample per launch) and, for the sum of the two launches it replaces, the plain
        // softmax kernel's chain time (`tail` mode: only this section - the answer to "what is in a 4.2 us launch": profiles/r05/timeline_tail.txt)
        float* prob = (float*)dalloc(8 * 1000 * 4, 0);
        unsigned* ctr = (unsigned*)dalloc(128, 0);
        run(P, "fc_i8_small + softmax 8x2048 -> 1000 (63 WGs)", 63, 7, [&] { launch_fc_i8_small_softmax(a, prob, ctr, P.st); });
        run(P, "softmax_f32 8 x 1000 (8 WGs; no stamps)", 8, 1, [&] { launch_softmax_rows(8, 1000, (const float*)a.y, prob, P.st); });
        run(P, "fc_i8_small, then softmax_f32 (two launches)", 63, 1, [&] { launch_fc_i8_small(a, P.st); launch_softmax_rows(8, 1000, (const float*)a.y, prob, P.st); });
        if (argc > 1 && !strcmp(argv[1], "tail")) return 0;
    }
    struct Img { const char* name; int N, HW, C, K, ib, rb, nw; };
    const Img imgs[] = {
        {"img3x3 res4 14x14x256 b8  1img x 7rows w4", 8, 14, 256, 256, 1, 7, 4},
        {"img3x3 res5 7x7x512 b8   1img x 7rows w4", 8, 7, 512, 512, 1, 7, 4},
        {"img3x3 res5 7x7x512 b8   2img x 7rows w4", 8, 7, 512, 512, 2, 7, 4},
        {"img3x3 res3 28x28x128 b8 1img x 7rows w4", 8, 28, 128, 128, 1, 7, 4},
        {"img3x3 res4 14x14x256 b1  1img x 7rows w4", 1, 14, 256, 256, 1, 7, 4},
        {"img3x3 res4 14x14x256 b1  1img x 2rows w4", 1, 14, 256, 256, 1, 2, 4},
    };
    // phases: 0 entry, 1 weights requested + table built, 2 DMA issued, 3 own DMA landed, 4 everyone's landed, 5 MFMAs done, 6 stored
    for (const Img& g : imgs) {
        if (stem_only) break;
        ConvKArgs a;
        memset(&a, 0, sizeof a);
        const int kg = 9 * g.C, kgp = (kg + 1023) / 1024 * 1024;
        a.N = g.N; a.H = a.W = a.OH = a.OW = g.HW; a.C = g.C; a.K = g.K; a.kh = a.kw = 3; a.pad_h = a.pad_w = 1;
        a.stride_h = a.stride_w = a.dil_h = a.dil_w = 1;
        a.M = g.N * g.HW * g.HW; a.Kg = kg; a.Kg_pad = kgp; a.in_u8 = 1; a.out_dtype = DT_U8; a.relu = 1; a.epi = EPI_I8_CONV;
        a.x = dalloc((size_t)a.M * g.C, -1); a.w = dalloc((size_t)((g.K + 127) / 128 * 128) * kgp, -1); a.zero = zero;
        a.y = dalloc((size_t)a.M * g.K, 0);
        a.scale = (const float*)dalloc(2048 * 4, 0); a.bias = (const float*)dalloc(2048 * 4, 0); a.comp = (const int*)dalloc(2048 * 4, 0);
        
        const int blocks = ((g.N + g.ib - 1) / g.ib) * ((g.HW + g.rb - 1) / g.rb) * ((g.K + 15) / 16);
        run(P, g.name, blocks, 7, [&] { launch_img_e1(a, g.nw, g.ib, g.rb, P.st); });
    }
    for (int nb : {8, 1}) {   // ---- conv1 7x7/2 + pool1 3x3/2 (f32 NCHW image in, u8 NHWC pooled out)
        // phases: 0 entry, 1 weights requested, 2 input patch staged (loads + quantise + LDS), 3 MFMAs done, 4 conv tile in LDS, 5 stored
        ConvKArgs a;
        memset(&a, 0, sizeof a);
        a.N = nb; a.H = a.W = 224; a.OH = a.OW = 112; a.C = 4; a.K = 64; a.kh = a.kw = 7; a.pad_h = a.pad_w = 3;
        a.stride_h = a.stride_w = 2; a.dil_h = a.dil_w = 1; a.kw_pad = 8; a.Kg = 7 * 8 * 4; a.Kg_pad = 256; a.Cin = 3; a.qinv = 50.f;
        a.M = nb * 112 * 112; a.out_dtype = DT_U8; a.relu = 1; a.epi = EPI_I8_CONV; a.pool_oh = a.pool_ow = 56;
        a.x = dalloc((size_t)nb * 3 * 224 * 224 * 4, 0); a.w = dalloc(128 * 256, -1); a.zero = zero; a.y = dalloc((size_t)nb * 56 * 56 * 64, 0);
        a.scale = (const float*)dalloc(2048 * 4, 0); a.bias = (const float*)dalloc(2048 * 4, 0);
        run(P, nb == 8 ? "stem 7x7/2 + maxpool 3x3/2, f32 image in, batch 8" : "stem 7x7/2 + maxpool 3x3/2, batch 1", nb * 98, 6,
            [&] { launch_conv_stem_pool(1, a, P.st); });
        // ... with the sibling pair reading the pooled tensor in the same launch (phase 6: the pair's outputs stored)
        StemPairKArgs ka;
        memset(&ka, 0, sizeof ka);
        ka.c = a;
        ka.c.y = nullptr;
        ka.t.w = dalloc(320 * 64, -1); ka.t.prm = dalloc(256 * 16, 0);
        ka.t.y1 = dalloc((size_t)nb * 56 * 56 * 256, 0); ka.t.y2 = dalloc((size_t)nb * 56 * 56 * 64, 0);
        ka.t.K1 = 256; ka.t.K2 = 64; ka.t.relu2 = 1; ka.t.u8_2 = 1;
        run(P, nb == 8 ? "stem + maxpool + pair 64 -> 256 | 64, batch 8" : "stem + maxpool + pair, batch 1", nb * 98, 7,
            [&] { launch_conv_stem_pool_pair(1, ka, P.st); });
    }
    if (stem_only) return 0;
    // ---- the implicit-GEMM / halo kernels on typical ResNet50 layers (batch 8 and batch 1) -----------------------------
    // phases: 0 entry, 1 gather state set up, 2 first stage in LDS (dma: ring prefetch issued), 3 reduction done, 4 stored
    struct Lay { const char* name; int N, HW, C, K, k, stride, pad, elt, kind, tile, ks, wg, early; };   // kind 0 reg, 1 dma, 2 halo(th=tile)
    const Lay lays[] = {
        {"res2 expand 64->256 @56 +eltwise  64x64 k1", 8, 56, 64, 256, 1, 1, 0, 1, 0, TILE_64x64, 1, 0, 0},
        {"res2 reduce 256->64 @56           64x128 k2", 8, 56, 256, 64, 1, 1, 0, 0, 0, TILE_64x128, 2, 0, 0},
        {"res2 3x3 64->64 @56               halo 4x16", 8, 56, 64, 64, 3, 1, 1, 0, 2, 4, 0, 0, 0},
        {"res3 expand 128->512 @28 +eltwise 64x64 k1", 8, 28, 128, 512, 1, 1, 0, 1, 0, TILE_64x64, 1, 0, 0},
        {"res3 3x3 128->128 @28             halo 4x16", 8, 28, 128, 128, 3, 1, 1, 0, 2, 4, 0, 0, 0},
        {"res4 reduce 1024->256 @14         64x32 k4 dma wg2", 8, 14, 1024, 256, 1, 1, 0, 0, 1, TILE_64x32, 4, 2, 0},
        {"res4 3x3 256->256 @14             64x32 k4 dma wg2", 8, 14, 256, 256, 3, 1, 1, 0, 1, TILE_64x32, 4, 2, 0},
        {"res4 expand 256->1024 @14 +elt    128x64 k4 reg", 8, 14, 256, 1024, 1, 1, 0, 1, 0, TILE_128x64, 4, 0, 0},
        {"res5 3x3 512->512 @7              32x32 k4 dma wg4", 8, 7, 512, 512, 3, 1, 1, 0, 1, TILE_32x32, 4, 4, 0},
        {"res2 expand 64->256 @56 +eltwise  64x64 k1 EARLY", 8, 56, 64, 256, 1, 1, 0, 1, 0, TILE_64x64, 1, 0, 1},
        {"res4 expand 256->1024 @14 +elt    128x64 k4 reg EARLY", 8, 14, 256, 1024, 1, 1, 0, 1, 0, TILE_128x64, 4, 0, 1},
        {"b1 res2 expand 64->256 @56 +elt   64x64 k1 EARLY", 1, 56, 64, 256, 1, 1, 0, 1, 0, TILE_64x64, 1, 0, 1},
        {"b1 res2 expand 64->256 @56 +elt   64x64 k1", 1, 56, 64, 256, 1, 1, 0, 1, 0, TILE_64x64, 1, 0, 0},
        {"b1 res4 3x3 256->256 @14          32x32 k4 dma wg4", 1, 14, 256, 256, 3, 1, 1, 0, 1, TILE_32x32, 4, 4, 0},
    };
    for (const Lay& g : lays) {
        ConvKArgs a;
        memset(&a, 0, sizeof a);
        const int OHW = (g.HW + 2 * g.pad - g.k) / g.stride + 1;
        const int kg = g.k * g.k * g.C, kgp = (kg + 1023) / 1024 * 1024;
        a.N = g.N; a.H = a.W = g.HW; a.OH = a.OW = OHW; a.C = g.C; a.K = g.K; a.kh = a.kw = g.k; a.pad_h = a.pad_w = g.pad;
        a.stride_h = a.stride_w = g.stride; a.dil_h = a.dil_w = 1;
        a.M = g.N * OHW * OHW; a.Kg = kg; a.Kg_pad = kgp; a.in_u8 = 1; a.epi = EPI_I8_CONV;
        a.inv_ohw = 1.f / (OHW * OHW); a.inv_ow = 1.f / OHW;
        a.x = dalloc((size_t)g.N * g.HW * g.HW * g.C, -1); a.w = dalloc((size_t)((g.K + 127) / 128 * 128) * kgp, -1); a.zero = zero;
        a.y = dalloc((size_t)a.M * g.K, 0);
        a.scale = (const float*)dalloc(2048 * 4, 0); a.bias = (const float*)dalloc(2048 * 4, 0); a.comp = (const int*)dalloc(2048 * 4, 0);
        if (g.elt) {
            a.out_dtype = DT_S8; a.relu = 0; a.res_mode = RES_ELTWISE; a.res_relu = 1; a.res = dalloc((size_t)a.M * g.K, -1);
            a.coeff_conv = a.coeff_res = 1.f; a.scale_conv = 0.05f; a.scale_res = 0.04f; a.res_dtype = DT_S8;
        } else {
            a.out_dtype = DT_U8; a.relu = 1;
        }
        int blocks;
        if (g.kind == 2) {
            blocks = g.N * ((OHW + 15) / 16) * ((OHW + g.tile - 1) / g.tile) * ((g.K + 63) / 64);
            run(P, g.name, blocks, 5, [&] { launch_halo_e1(g.tile, a, P.st); });
        } else {
            int bmk, bnp;
            tile_dims(g.tile, &bmk, &bnp);
            blocks = ((a.M + bnp - 1) / bnp) * ((g.K + bmk - 1) / bmk);
            const int estage = 64 * g.ks * (g.wg > 1 ? g.wg : 1);
            a.steps = (kg + estage - 1) / estage;
            if (g.kind == 1) run(P, g.name, blocks, 6, [&] { launch_igemm_dma_m0_e1(g.tile, g.ks, g.wg, a, P.st); });
            else if (g.elt) run(P, g.name, blocks, 5, [&] { launch_igemm_m0_e2(g.tile, g.ks, a, P.st); });
            else run(P, g.name, blocks, 5, [&] { launch_igemm_m0_e1(g.tile, g.ks, a, P.st); });
        }
    }
    // ---- FP32 on three bf16 planes (MODE 3), ResNet50 deep-K layers at batch 8, unsplit and split-K ----------------------
    // phases: 0 entry, 1 gather state set up, 2 first stage in LDS, 5 reduction loop done, 3 split-K hand-off done, 4 stored
    struct L3 { const char* name; int N, HW, C, K, k, tile, ks, sh; };
    const L3 l3[] = {
        {"f32 bf16x3 res4 3x3 256->256 @14  64x64 k1", 8, 14, 256, 256, 3, TILE_64x64, 1, 0},
        {"f32 bf16x3 res4 3x3 256->256 @14  64x64 k1 split4", 8, 14, 256, 256, 3, TILE_64x64, 1, 2},
        {"f32 bf16x3 res4 3x3 256->256 @14  32x32 k2", 8, 14, 256, 256, 3, TILE_32x32, 2, 0},
        {"f32 bf16x3 res5 3x3 512->512 @7   64x64 k1 split8", 8, 7, 512, 512, 3, TILE_64x64, 1, 3},
        {"f32 bf16x3 res3 3x3 128->128 @28  64x64 k2", 8, 28, 128, 128, 3, TILE_64x64, 2, 0},
    };
    for (const L3& g : l3) {
        ConvKArgs a;
        memset(&a, 0, sizeof a);
        const int pad = g.k / 2, OHW = g.HW;
        const int kg = g.k * g.k * g.C, kgp = (kg + 63) / 64 * 64, kpad = (g.K + 127) / 128 * 128;
        a.N = g.N; a.H = a.W = g.HW; a.OH = a.OW = OHW; a.C = g.C; a.K = g.K; a.kh = a.kw = g.k; a.pad_h = a.pad_w = pad;
        a.stride_h = a.stride_w = 1; a.dil_h = a.dil_w = 1;
        a.M = g.N * OHW * OHW; a.Kg = kg; a.Kg_pad = kgp; a.epi = EPI_F32; a.out_dtype = DT_F32; a.relu = 1;
        a.inv_ohw = 1.f / (OHW * OHW); a.inv_ow = 1.f / OHW;
        a.x = dalloc((size_t)g.N * g.HW * g.HW * g.C * 4, 0); a.w = dalloc((size_t)3 * kpad * kgp * 2 + 65536, 0); a.zero = zero;
        a.w_plane_chunks = (int)((size_t)kpad * kgp / 8);
        a.y = dalloc((size_t)a.M * g.K * 4, 0);
        a.bias = (const float*)dalloc(2048 * 4, 0);
        a.ksplit_sh = g.sh;
        a.part = (float*)dalloc(((size_t)a.M + 127) * (g.K + 127) * 4 * 8, 0);
        a.part_ctr = (unsigned*)dalloc(65536, 0);
        int bmk, bnp;
        tile_dims(g.tile, &bmk, &bnp);
        const int tiles = ((a.M + bnp - 1) / bnp) * ((g.K + bmk - 1) / bmk);
        const int blocks = g.sh ? (8 * ((tiles + 7) / 8)) << g.sh : tiles;
        a.steps = (kg + 32 * g.ks - 1) / (32 * g.ks);
        run(P, g.name, blocks, 6, [&] { launch_igemm_m3_e3(g.tile, g.ks, a, P.st); });
    }
    return 0;
}
