// Developer probe: the DPP/permlane tree must give lane 0 the same bits as the shuffle tree.
#include "../../lrf_amd/csrc/lrf_kernels.hip"
__device__ double tree_ref(double v)
{
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_down(v, off, 64);
    return __shfl(v, 0, 64);
}
__global__ void k_tree(const double* in, double* out)
{
    double v = in[blockIdx.x * 64 + threadIdx.x];
    double a = wave_tree64(v), b = tree_ref(v);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = a; out[2 * blockIdx.x + 1] = b; }
}
extern "C" int tree_run(const double* h_in, double* h_out, int nblk)
{
    double *d_in, *d_out;
    if (hipMalloc(&d_in, nblk * 64 * 8) != hipSuccess || hipMalloc(&d_out, nblk * 2 * 8) != hipSuccess) return -1;
    (void)hipMemcpy(d_in, h_in, nblk * 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_tree, dim3(nblk), dim3(64), 0, 0, d_in, d_out);
    (void)hipMemcpy(h_out, d_out, nblk * 2 * 8, hipMemcpyDeviceToHost);
    return 0;
}
