// Input step of the evaluation path on the GPU: the reference pads every test pair on the CPU
// (`pad_to_2x`, datasets/data_augmentation/__init__.py:57-80: top / right zero padding of the uint8 HWC images to
// multiples of 96) and then normalises it (`get_transform`, datasets/utils.py:62-69: torchvision ToTensor = HWC uint8
// -> CHW float / 255, Normalize = (x - mean) / std with the ImageNet statistics) -- three passes over the image in
// numpy / torchvision plus the host->device copy of the 4x larger float tensor.  Here the uint8 image is copied to the
// device as it is (3 B/pixel) and one kernel writes the padded, normalised NCHW float tensor.
//
//   out[b][c][y][x] = ((y >= top && x < W ? img[b][y - top][x][c] : 0) / 255 - mean[c]) / std[c]
//
// (padded pixels are black BEFORE normalisation, i.e. -mean/std after it, exactly as in the reference where the padding
// precedes the transform).  The arithmetic is the reference's, operation for operation: an IEEE fp32 division by 255,
// a subtraction and a division by std -- bit-identical to torchvision's `div(255)`, `sub_(mean).div_(std)`.
// Roofline: HBM, algorithmic bytes = 3 B read + 12 B written per padded pixel.  One thread = 4 consecutive pixels of one
// row: 12 bytes in, three 16-byte stores out (one per colour plane).
#include "stx_common.h"

namespace {

constexpr int PP_THREADS = 256;

struct PpArgs {
    const unsigned char* img;   // [B][H][W][3]
    float* out;                 // [B][3][Hp][Wp]
    int H, W, Hp, Wp, top;
    float mean[3], std[3];
};

__global__ __launch_bounds__(PP_THREADS) void pad_normalize_u8_kernel(PpArgs a) {
    const int q = blockIdx.x * PP_THREADS + threadIdx.x;           // quad index within the padded image
    const int b = blockIdx.y;
    const int wq = a.Wp >> 2;
    if (q >= a.Hp * wq) return;
    const int y = q / wq, x0 = (q - y * wq) * 4;
    const int sy = y - a.top;
    float v[3][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int x = x0 + p;
        const bool in = sy >= 0 && x < a.W;
        const unsigned char* px = a.img + (((size_t)b * a.H + (in ? sy : 0)) * a.W + (in ? x : 0)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float u = in ? (float)px[c] : 0.f;
            v[c][p] = (u / 255.0f - a.mean[c]) / a.std[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        stx_st4(a.out + (((size_t)b * 3 + c) * a.Hp + y) * a.Wp + x0, make_float4(v[c][0], v[c][1], v[c][2], v[c][3]));
}

}  // namespace

extern "C" int stx_pad_normalize_u8(const unsigned char* img, float* out, int B, int H, int W, int Hp, int Wp, int top,
                                    const float* mean3, const float* std3, void* stream) {
    stx_begin();
    STX_REQUIRE(img && out && mean3 && std3 && B > 0 && H > 0 && W > 0, "pad_normalize_u8: bad arguments");
    STX_REQUIRE(Hp >= H && Wp >= W && top == Hp - H, "pad_normalize_u8: the padding goes to the top and to the right (top = Hp - H)");
    STX_REQUIRE(Wp % 4 == 0, "pad_normalize_u8: padded width %d must be a multiple of 4", Wp);
    PpArgs a;
    a.img = img; a.out = out; a.H = H; a.W = W; a.Hp = Hp; a.Wp = Wp; a.top = top;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.std[c] = std3[c]; }
    hipLaunchKernelGGL(pad_normalize_u8_kernel, dim3(stx_cdiv(Hp * (Wp / 4), PP_THREADS), B), dim3(PP_THREADS), 0,
                       (hipStream_t)stream, a);
    return stx_check_launch("pad_normalize_u8");
}
