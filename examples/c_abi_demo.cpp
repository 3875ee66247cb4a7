// The C ABI without Python: what a maintainer binding libkeep_hip.so from C / C++ / Go / Java writes.
//
//   hipcc -O2 -Iinclude examples/c_abi_demo.cpp -Lkeep_amd -lkeep_hip -Wl,-rpath,$PWD/keep_amd -o examples/c_abi_demo
//   python tools/dump_state_dict.py weights.bin tiles.bin        # seeded synthetic release-layout state_dict + bf16 tiles
//   examples/c_abi_demo weights.bin tiles.bin out.bin            # -> fp32 [B,768] embeddings
//
// weights.bin: records {u32 key_len, key bytes, u32 ndim, i64 shape[ndim], f32 data[prod(shape)]} until EOF.
// tiles.bin:   {i64 B, i32 pix_dtype, raw pixels [B,3,224,224]}  (KEEP_PIX_* codes of include/keep_hip.h).
// Replaces, line for line, quick_start/keep_inference.py:79-85 (build + load_state_dict + eval) and :101 (encode_image).
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "keep_hip.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)

int main(int argc, char** argv) {
    if (argc != 4) DIE("usage: %s weights.bin tiles.bin out.bin", argv[0]);
    keep_handle* h = nullptr;
    if (keep_create(0, &h)) DIE("keep_create failed (no MI355X?)");

    // load_state_dict(strict=True): one keep_load_tensor per entry (host pointers), then finalize
    FILE* f = fopen(argv[1], "rb");
    if (!f) DIE("cannot open %s", argv[1]);
    uint32_t klen;
    size_t n_tensors = 0;
    while (fread(&klen, 4, 1, f) == 1) {
        std::string key(klen, '\0');
        uint32_t ndim;
        if (fread(&key[0], 1, klen, f) != klen || fread(&ndim, 4, 1, f) != 1) DIE("truncated record");
        std::vector<int64_t> shape(ndim ? ndim : 1, 1);
        if (ndim && fread(shape.data(), 8, ndim, f) != ndim) DIE("truncated shape");
        size_t numel = 1;
        for (uint32_t i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
        std::vector<float> data(numel);
        if (fread(data.data(), 4, numel, f) != numel) DIE("truncated data for %s", key.c_str());
        if (keep_load_tensor(h, key.c_str(), data.data(), (int)ndim, shape.data(), /*on_device=*/0))
            DIE("keep_load_tensor(%s): %s", key.c_str(), keep_last_error(h));
        ++n_tensors;
    }
    fclose(f);
    if (keep_finalize_weights(h)) DIE("keep_finalize_weights: %s", keep_last_error(h));     // missing key -> KEEP_EKEY
    fprintf(stderr, "loaded %zu tensors: ViT depth %d, BERT layers %d\n", n_tensors, keep_vit_depth(h), keep_bert_layers(h));

    // encode_image: device buffers are the caller's
    f = fopen(argv[2], "rb");
    if (!f) DIE("cannot open %s", argv[2]);
    int64_t B; int32_t dtype;
    if (fread(&B, 8, 1, f) != 1 || fread(&dtype, 4, 1, f) != 1) DIE("bad tiles header");
    const size_t px = dtype == KEEP_PIX_F32 ? 4 : (dtype == KEEP_PIX_U8_HWC ? 1 : 2);
    const size_t in_bytes = (size_t)B * 3 * 224 * 224 * px, out_bytes = (size_t)B * 768 * sizeof(float);
    std::vector<unsigned char> pixels(in_bytes);
    if (fread(pixels.data(), 1, in_bytes, f) != in_bytes) DIE("truncated pixels");
    fclose(f);
    void* d_in = nullptr; float* d_out = nullptr;
    hipStream_t stream;
    if (hipMalloc(&d_in, in_bytes) || hipMalloc((void**)&d_out, out_bytes) || hipStreamCreate(&stream)) DIE("hipMalloc failed");
    if (keep_reserve(h, B, 0, 0)) DIE("keep_reserve: %s", keep_last_error(h));              // no allocation inside the hot call
    if (hipMemcpyAsync(d_in, pixels.data(), in_bytes, hipMemcpyHostToDevice, stream)) DIE("H2D copy failed");
    if (keep_encode_image(h, d_in, dtype, B, d_out, stream)) DIE("keep_encode_image: %s", keep_last_error(h));
    std::vector<float> out((size_t)B * 768);
    if (hipMemcpyAsync(out.data(), d_out, out_bytes, hipMemcpyDeviceToHost, stream)) DIE("D2H copy failed");
    if (hipStreamSynchronize(stream) != hipSuccess) DIE("stream failed");

    f = fopen(argv[3], "wb");
    if (!f || fwrite(out.data(), 4, out.size(), f) != out.size()) DIE("cannot write %s", argv[3]);
    fclose(f);
    double norm0 = 0;
    for (int i = 0; i < 768; ++i) norm0 += (double)out[i] * out[i];
    fprintf(stderr, "encoded %lld tiles; |embedding 0|^2 = %.6f (unit norm expected)\n", (long long)B, norm0);
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipStreamDestroy(stream);
    keep_destroy(h);
    return 0;
}
