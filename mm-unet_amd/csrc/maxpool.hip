// maxpool.hip -- backward of nn.MaxPool2d(kernel_size=3, stride=2, padding=1) as a gather.
//
// Where it sits: MM_Net's stem (src/UM_Net/MMUNet.py:493,537 `self.maxpool`, applied to the 64 x 256 x 256 stem map).  The
// forward stays ATen's (0.09 ms; it returns the arg-max indices); ATen's backward zero-fills d input and scatters the
// output gradients with float atomics (windows overlap): 18 + 221 us for [8, 64, 128, 128] -> [8, 64, 256, 256].
// Here every input pixel looks at the (at most four) windows that contain it and adds the gradients of those whose
// recorded arg-max is this pixel: no atomics, no zero fill, bit-reproducible; 134 MB written + 100 MB read.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float *__restrict__ g, const long *__restrict__ idx,
                                                             float *__restrict__ dx, int H, int W, int OH, int OW,
                                                             long planes) {
    const int wq = (W + 3) / 4;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= planes * H * wq) return;
    const int q = (int)(t % wq);
    const long r = t / wq;
    const int y = (int)(r % H);
    const long plane = r / H;
    const float *gp = g + plane * OH * OW;
    const long *ip = idx + plane * OH * OW;
    // windows (oy, ox) cover rows 2 oy - 1 .. 2 oy + 1: input row y lies in oy = y / 2 and, for odd y, (y + 1) / 2
    const int oy0 = y >> 1, oy1 = (y + 1) >> 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = q * 4 + j;
        if (x >= W) break;
        const long me = (long)y * W + x;
        const int ox0 = x >> 1, ox1 = (x + 1) >> 1;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = a == 0 ? oy0 : oy1;
            if ((a == 1 && oy1 == oy0) || oy >= OH) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ox = b == 0 ? ox0 : ox1;
                if ((b == 1 && ox1 == ox0) || ox >= OW) continue;
                const long o = (long)oy * OW + ox;
                if (ip[o] == me) acc[j] += gp[o];
            }
        }
    }
    float *dst = dx + (plane * H + y) * W + q * 4;
    if ((W & 3) == 0) {
        *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
        for (int j = 0; j < 4 && q * 4 + j < W; ++j) dst[j] = acc[j];
    }
}

}  // namespace

extern "C" int mmu_maxpool3s2_bwd(const mmu_maxpool_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "maxpool3s2_bwd: null params");
    MMU_CHECK(p->planes > 0 && p->height > 0 && p->width > 0, "maxpool3s2_bwd: empty tensor");
    MMU_CHECK(p->out_height == (p->height - 1) / 2 + 1 && p->out_width == (p->width - 1) / 2 + 1,
              "maxpool3s2_bwd: output size must be that of kernel 3 / stride 2 / padding 1 (got %d x %d for %d x %d)",
              p->out_height, p->out_width, p->height, p->width);
    MMU_CHECK(p->dout && p->indices && p->dinput, "maxpool3s2_bwd: dout, indices, dinput are required");
    MMU_CHECK(((uintptr_t)p->dinput & 15) == 0, "maxpool3s2_bwd: dinput must be 16-byte aligned");
    const long total = (long)p->planes * p->height * ((p->width + 3) / 4);
    MMU_CHECK(total < (1L << 40), "maxpool3s2_bwd: tensor too large");
    maxpool3s2_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        p->dout, (const long *)p->indices, p->dinput, p->height, p->width, p->out_height, p->out_width, p->planes);
    MMU_HIP_LAUNCH_CHECK("maxpool3s2_bwd");
    return 0;
}
