// Host-logic check of the C-ABI under AddressSanitizer + UndefinedBehaviorSanitizer.
// Built by `make -C crfp_amd/csrc asan` from the library's sources compiled HOST-ONLY (hipcc --cuda-host-only -fsanitize=address,undefined):
// no device code, no GPU -- this runs in the CPU build container, never on the GPU box.  It drives what the library does on the host before
// any kernel is launched: workspace / packed-weight sizing (the Layout arenas of engine.hip / engine_rt.hip over many geometries and batch
// sizes), the parameter tables, and every argument-error path (tests/test_host_logic.py::test_argument_errors_do_not_touch_the_gpu,
// test_sizes_are_sane).  Exit code 0 = every expectation held and no sanitizer report was raised.
#include "../../include/crfp_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <cstdint>
extern "C" long crfp_stub_launches(void);

static int fails = 0;
#define EXPECT(cond)                                                              \
    do {                                                                          \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++fails; } \
    } while (0)

int main() {
    EXPECT(crfp_version() > 0);
    // ---- sizes (test_sizes_are_sane) over geometries incl. ragged and degenerate ones
    const size_t pw = crfp_dsv_packed_weight_bytes(0);
    EXPECT(pw > 9000000 && pw < 40000000);
    EXPECT(crfp_dsv_packed_weight_bytes(1) > 0 && crfp_dsv_packed_weight_bytes_bf16(0) > 0 && crfp_cra_packed_weight_bytes(0) > pw / 2);
    const size_t a = crfp_dsv_workspace_bytes(7, 180, 320), b = crfp_dsv_workspace_bytes(7, 270, 480);
    EXPECT(a > 1000000000ull && a < 4000000000ull && b > 2 * a && b < 5 * a / 2);
    EXPECT(crfp_dsv_workspace_bytes(0, 180, 320) == 0 && crfp_dsv_workspace_bytes(7, 4, 4) == 0);
    EXPECT(crfp_flow_warp_workspace_bytes(1, 32, 360, 640) == 2ull * 32 * 360 * 640 * 4);
    const int hs[] = {8, 9, 17, 20, 33, 64, 101, 180, 270}, ws[] = {8, 16, 26, 36, 47, 65, 130, 170, 320, 480};
    for (int h : hs)
        for (int w : ws)
            for (int t : {1, 2, 7, 33, 100}) {
                const size_t s1 = crfp_dsv_workspace_bytes(t, h, w), sb = crfp_dsv_workspace_bytes_bf16(t, h, w);
                EXPECT(s1 > 0 && sb > 0 && sb <= s1);
                EXPECT(crfp_dsv_status_offset(t, h, w) < s1 && crfp_dsv_status_offset_bf16(t, h, w) < sb);
                for (int n : {1, 2, 4, 65}) {
                    const size_t sn = crfp_dsv_batch_workspace_bytes(n, t, h, w);
                    EXPECT(sn >= s1 / 2 && crfp_dsv_batch_status_offset(n, t, h, w) + 4ull * n <= sn);
                    EXPECT(crfp_dsv_batch_workspace_bytes_bf16(n, t, h, w) > 0 && crfp_cra_batch_workspace_bytes(n, t, h, w) >= sn / 2);
                    EXPECT(crfp_cra_batch_status_offset(n, t, h, w) < crfp_cra_batch_workspace_bytes(n, t, h, w));
                    EXPECT(crfp_simple_batch_workspace_bytes(n, t, h, w) > sn && crfp_dense_batch_workspace_bytes_bf16(n, t, h, w) > 0);   // 8 feature quads per level instead of 6
                    EXPECT(crfp_simple_batch_status_offset(n, t, h, w) + 4ull * n <= crfp_simple_batch_workspace_bytes(n, t, h, w));
                }
            }
    EXPECT(crfp_dsv_batch_workspace_bytes(0, 7, 180, 320) == 0 && crfp_dsv_batch_workspace_bytes(-3, 7, 180, 320) == 0);
    EXPECT(crfp_conv3x3_workspace_bytes(1, 64, 32, 360, 640) > 0 && crfp_conv3x3_packed_bytes(64, 32) > 0 && crfp_dcnv2_g8_packed_bytes() > 0);
    EXPECT(crfp_conv3x3_ex_workspace_bytes(1, 32, 32, 32, 64, 64, 0, 2, 1) > 0);
    EXPECT(crfp_dcnv2_workspace_bytes(1, 32, 32, 360, 640, 3, 8) > 0 && crfp_dcnv2_shared_workspace_bytes(1, 4, 1440, 2560) > 0);
    EXPECT(crfp_spynet_workspace_bytes(1, 180, 320) > 0 && crfp_fovea_head_workspace_bytes(1, 1440, 2560) > 0);
    EXPECT(crfp_rt_packed_weight_bytes(0) > 0);
    EXPECT(crfp_rt_workspace_bytes(5, 135, 240, 96, 96, 720, 720) > 0);      // test_runtime.py's geometry
    EXPECT(crfp_rt_workspace_bytes(5, 135, 240, 2000, 96, 720, 720) == 0);   // fovea larger than the frame: bad geometry, reported
    EXPECT(crfp_rt_workspace_bytes(5, 135, 240, 96, 96, 724, 720) == 0);     // warp_size no multiple of 8
    // ---- the parameter tables
    int np = 0;
    while (crfp_dsv_param_name(np)) { EXPECT(crfp_dsv_param_numel(np, 0) > 0 && std::strlen(crfp_dsv_param_name(np)) > 3); ++np; }
    EXPECT(np == 118 && crfp_dsv_param_name(-1) == nullptr && crfp_dsv_param_numel(np, 0) <= 0);
    int nc = 0;
    while (crfp_cra_param_name(nc)) { EXPECT(crfp_cra_param_numel(nc, 0) > 0); ++nc; }
    EXPECT(nc > np);
    // CRFP_simple / CRFP: CRFP_DSV's names, four other shapes (upsample: 128 rows; upsample_post: 32 columns; dense: 96 / 12 columns in main.0)
    {
        int other_s = 0, other_d = 0;
        for (int i = 0; i < np; ++i) {
            EXPECT(crfp_simple_param_numel(i, 0) > 0 && crfp_dense_param_numel(i, 1) > 0);
            other_s += crfp_simple_param_numel(i, 0) != crfp_dsv_param_numel(i, 0);
            other_d += crfp_dense_param_numel(i, 0) != crfp_dsv_param_numel(i, 0);
        }
        EXPECT(other_s == 3 && other_d == 7 && crfp_simple_param_numel(np, 0) <= 0 && crfp_dense_param_numel(-1, 0) <= 0);
    }
    int nr = 0;
    while (crfp_rt_param_name(nr)) { EXPECT(crfp_rt_param_numel(nr, 0) > 0); ++nr; }
    EXPECT(nr > 0);
    // ---- argument errors: reported before anything touches the GPU (test_argument_errors_do_not_touch_the_gpu)
    float* const p16 = reinterpret_cast<float*>(16);   // a non-null, never dereferenced host-side pointer value
    EXPECT(crfp_dsv_forward_clip(nullptr, 0, nullptr, nullptr, nullptr, nullptr, 7, 180, 320, nullptr, 0, nullptr) == -1);
    EXPECT(std::strstr(crfp_last_error_string(), "null") != nullptr);
    EXPECT(crfp_flow_warp_f32(nullptr, nullptr, nullptr, 1, 4, 8, 8, 0, nullptr, 0, nullptr) == -1);
    EXPECT(crfp_dcnv2_forward_f32(p16, p16, p16, p16, p16, p16, 1, 32, 32, 8, 8, 5, 2, 1, 8, nullptr, 0, nullptr) == -3);
    EXPECT(std::strstr(crfp_last_error_string(), "kernel 3") != nullptr);
    EXPECT(crfp_dcnv2_forward_f32(p16, p16, p16, p16, p16, p16, 1, 30, 32, 8, 8, 3, 1, 1, 8, nullptr, 0, nullptr) == -1);
    EXPECT(crfp_conv3x3_f32(p16, p16, p16, p16, 1, 3, 8, 8, 8, 9, 1.0f, nullptr, 0, nullptr) == -1);
    // too small a workspace / bad dimensions on the engines: refused by the host-side checks
    const unsigned char* const m16 = reinterpret_cast<const unsigned char*>(16);
    EXPECT(crfp_dsv_forward_clip(p16, 0, p16, p16, m16, p16, 7, 180, 320, p16, 1024, nullptr) != 0);
    EXPECT(crfp_dsv_forward_clip(p16, 0, p16, p16, m16, p16, 0, 180, 320, p16, (size_t)1 << 40, nullptr) != 0);
    EXPECT(crfp_dsv_forward_batch(p16, 0, p16, p16, m16, p16, 0, 7, 180, 320, p16, (size_t)1 << 40, nullptr) != 0);
    EXPECT(crfp_dsv_forward_batch_bf16(p16, 0, p16, p16, m16, p16, 2, 7, 180, 320, p16, 64, nullptr) != 0);
    EXPECT(crfp_cra_forward_batch(p16, 0, p16, p16, m16, p16, 2, 7, 180, 320, p16, 64, nullptr) != 0);
    EXPECT(crfp_simple_forward_batch(p16, 0, p16, p16, m16, p16, 2, 7, 180, 320, p16, 64, nullptr) != 0);
    EXPECT(crfp_dense_forward_batch_bf16(p16, 0, p16, nullptr, m16, p16, 1, 7, 180, 320, p16, (size_t)1 << 40, nullptr) != 0);
    EXPECT(crfp_dsv_forward_clip_bf16(p16, CRFP_DSV_STRICT_F32, p16, p16, m16, p16, 7, 180, 320, p16, (size_t)1 << 40, nullptr) != 0);
    EXPECT(std::strlen(crfp_last_error_string()) > 0);
    // ---- whole engine calls on the stub runtime (tools/asan_host/hip_stub.cpp: every HIP call succeeds, no kernel runs): the host side of a
    // call -- argument checks, Layout arenas, the launch-argument tables of ~50 launches per frame, the fork / join of the side stream, the
    // per-thread stream table and crfp_shutdown() -- under the sanitizers.  Device pointers are fabricated and never dereferenced on the host.
    {
        char* const dev = reinterpret_cast<char*>(0x100000000000ull);
        float* const lrs = reinterpret_cast<float*>(dev), *const fvs = reinterpret_cast<float*>(dev + (1ull << 36));
        const uint8_t* const mks = reinterpret_cast<const uint8_t*>(dev + (2ull << 36));
        float* const out = reinterpret_cast<float*>(dev + (3ull << 36));
        void* const wsp = dev + (4ull << 36);
        void* const pk = dev + (5ull << 36);
        void* const stream = nullptr;
        const float* params[256];
        for (int i = 0; i < 256; ++i) params[i] = reinterpret_cast<const float*>(dev + (6ull << 36) + ((size_t)i << 24));
        const long l0 = crfp_stub_launches();
        EXPECT(crfp_dsv_pack_weights(params, 0, pk, crfp_dsv_packed_weight_bytes(0), stream) == 0);
        EXPECT(crfp_dsv_pack_weights_bf16(params, 0, pk, crfp_dsv_packed_weight_bytes_bf16(0), stream) == 0);
        EXPECT(crfp_cra_pack_weights(params, 0, pk, crfp_cra_packed_weight_bytes(0), stream) == 0);
        EXPECT(crfp_simple_pack_weights(params, 1, pk, crfp_simple_packed_weight_bytes(1), stream) == 0);
        EXPECT(crfp_dense_pack_weights_bf16(params, 0, pk, crfp_dense_packed_weight_bytes_bf16(0), stream) == 0);
        EXPECT(crfp_dense_pack_weights(params, 0, pk, crfp_dsv_packed_weight_bytes(0), stream) != 0);   // the wider main.0 images need more
        EXPECT(crfp_rt_pack_weights(params, 0, pk, crfp_rt_packed_weight_bytes(0), stream) == 0);
        EXPECT(crfp_dsv_pack_weights(params, 0, pk, 1024, stream) != 0);   // packed buffer too small
        const int geo[][2] = {{20, 36}, {33, 47}, {101, 170}, {180, 320}};
        for (const auto& g : geo) {
            const int h = g[0], w = g[1];
            for (int t : {1, 3, 9}) {
                EXPECT(crfp_dsv_forward_clip(pk, 0, lrs, fvs, mks, out, t, h, w, wsp, crfp_dsv_workspace_bytes(t, h, w), stream) == 0);
                EXPECT(crfp_dsv_forward_clip(pk, CRFP_DSV_SINGLE_STREAM | CRFP_DSV_Y_ONLY, lrs, fvs, mks, out, t, h, w, wsp, crfp_dsv_workspace_bytes(t, h, w), stream) == 0);
                EXPECT(crfp_dsv_forward_clip(pk, CRFP_DSV_STRICT_F32, lrs, fvs, mks, out, t, h, w, wsp, crfp_dsv_workspace_bytes(t, h, w), stream) == 0);
                EXPECT(crfp_dsv_forward_clip_bf16(pk, 0, lrs, fvs, mks, out, t, h, w, wsp, crfp_dsv_workspace_bytes_bf16(t, h, w), stream) == 0);
                for (int n : {2, 4}) {
                    EXPECT(crfp_dsv_forward_batch(pk, 0, lrs, fvs, mks, out, n, t, h, w, wsp, crfp_dsv_batch_workspace_bytes(n, t, h, w), stream) == 0);
                    EXPECT(crfp_dsv_forward_batch_bf16(pk, 0, lrs, fvs, mks, out, n, t, h, w, wsp, crfp_dsv_batch_workspace_bytes_bf16(n, t, h, w), stream) == 0);
                    EXPECT(crfp_cra_forward_batch(pk, 0, lrs, fvs, mks, out, n, t, h, w, wsp, crfp_cra_batch_workspace_bytes(n, t, h, w), stream) == 0);
                    EXPECT(crfp_simple_forward_batch(pk, n == 2 ? CRFP_DSV_STRICT_F32 : 0, lrs, fvs, mks, out, n, t, h, w, wsp, crfp_simple_batch_workspace_bytes(n, t, h, w), stream) == 0);
                    EXPECT(crfp_dense_forward_batch(pk, CRFP_DSV_SINGLE_STREAM, lrs, fvs, mks, out, n, t, h, w, wsp, crfp_dense_batch_workspace_bytes(n, t, h, w), stream) == 0);
                    EXPECT(crfp_dense_forward_batch_bf16(pk, 0, lrs, fvs, mks, out, n, t, h, w, wsp, crfp_dense_batch_workspace_bytes_bf16(n, t, h, w), stream) == 0);
                    EXPECT(crfp_simple_forward_batch(pk, 0, lrs, fvs, mks, out, n, t, h, w, wsp, crfp_dsv_batch_workspace_bytes(n, t, h, w), stream) != 0);   // CRFP_DSV's smaller workspace
                    EXPECT(crfp_dsv_forward_batch(pk, 0, lrs, fvs, mks, out, n, t, h, w, wsp, crfp_dsv_batch_workspace_bytes(n, t, h, w) - 1, stream) != 0);
                }
            }
            // one frame per call: first call, then steady calls with and without the resident-inputs promise, with a regional mask, in a batch
            const size_t sws = crfp_dsv_batch_workspace_bytes(1, 1, h, w);
            EXPECT(crfp_dsv_stream_frame(pk, 0, lrs, nullptr, fvs, mks, nullptr, out, 1, h, w, wsp, sws, stream) == 0);
            EXPECT(crfp_dsv_stream_frame(pk, 0, lrs, lrs, fvs, mks, nullptr, out, 0, h, w, wsp, sws, stream) == 0);
            EXPECT(crfp_dsv_stream_frame(pk, 0, lrs, lrs, fvs, mks, mks, out, 0, h, w, wsp, sws, stream) == 0);
            EXPECT(crfp_dsv_stream_frame(pk, CRFP_DSV_INPUTS_RESIDENT, lrs, nullptr, fvs, mks, nullptr, out, 1, h, w, wsp, sws, stream) == 0);
            for (int k = 0; k < 3; ++k)
                EXPECT(crfp_dsv_stream_frame(pk, CRFP_DSV_INPUTS_RESIDENT, lrs, nullptr, fvs, mks, nullptr, out, 0, h, w, wsp, sws, stream) == 0);
            EXPECT(crfp_dsv_stream_frame_bf16(pk, 0, lrs, lrs, fvs, mks, nullptr, out, 0, h, w, wsp, crfp_dsv_batch_workspace_bytes_bf16(1, 1, h, w), stream) == 0);
            EXPECT(crfp_dsv_stream_batch(pk, 0, lrs, lrs, fvs, mks, nullptr, out, 0, 3, h, w, wsp, crfp_dsv_batch_workspace_bytes(3, 1, h, w), stream) == 0);
            EXPECT(crfp_dsv_stream_batch(pk, 0, lrs, lrs, fvs, mks, mks, out, 0, 3, h, w, wsp, crfp_dsv_batch_workspace_bytes(3, 1, h, w), stream) != 0);   // fg needs n = 1
            EXPECT(crfp_dsv_stream_batch(pk, 0, lrs, lrs, fvs, mks, nullptr, out, 0, 33, h, w, wsp, (size_t)1 << 44, stream) != 0);                          // n <= 32
        }
        EXPECT(crfp_rt_forward_clip(pk, 0, lrs, fvs, out, 5, 135, 240, 96, 96, 720, 720, wsp, crfp_rt_workspace_bytes(5, 135, 240, 96, 96, 720, 720), stream) == 0);
        EXPECT(crfp_rt_forward_clip(pk, CRFP_DSV_STRICT_F32, lrs, fvs, out, 5, 135, 240, 96, 96, 720, 720, wsp, (size_t)1 << 40, stream) != 0);
        EXPECT(crfp_fnet_forward(pk, lrs, lrs, out, 2, 180, 320, wsp, crfp_dsv_workspace_bytes(3, 180, 320), stream) == 0);
        EXPECT(crfp_debug_side_tables() >= 1);
        EXPECT(crfp_shutdown() == 0);
        EXPECT(crfp_dsv_forward_clip(pk, 0, lrs, fvs, mks, out, 2, 20, 36, wsp, crfp_dsv_workspace_bytes(2, 20, 36), stream) == 0);   // and a fresh call after it
        std::printf("host_check: %ld launches enqueued on the stub runtime\n", crfp_stub_launches() - l0);
    }
    std::printf("host_check: %d parameters (dsv) / %d (cra) / %d (rt), %d failed expectations\n", np, nc, nr, fails);
    return fails ? 1 : 0;
}
