// fetch_calibrate — known-byte kernels in the access patterns of this library, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on
// gfx950 (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane)
// ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   cal_stream16     16 bytes per lane, coalesced, every byte of a 512 MiB buffer once (the guide's case: expect counter = bytes / 2)
//   cal_stream8       8 bytes per lane, coalesced, every byte once (the staging loads of pp_k_plan_skips, track rows)
//   cal_gather384     8 bytes per lane, lanes 384 bytes apart, six fields per record (one lane per PPEdgeSetup: solve / approach)
//   cal_write8        8 bytes per lane coalesced stores, every byte once
//   cal_scatter384    8-byte stores, lanes 384 bytes apart, six fields per record (pp_k_solve_edges writing its records)
// build + run: tools/fetch_calibrate.sh  (hipcc, then one rocprofv3 --pmc pass per counter)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void cal_stream16(const double2* p, size_t n, double* sink) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 12345.678) *sink = acc;
}
__global__ void cal_stream8(const double* p, size_t n, double* sink) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.678) *sink = acc;
}
__global__ void cal_gather384(const double* p, size_t nrec, double* sink) {      // record = 48 doubles; fields 0, 9, 17, 30, 37, 43
    double acc = 0;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += (size_t)gridDim.x * blockDim.x) {
        const double* q = p + r * 48;
        acc += q[0] + q[9] + q[17] + q[30] + q[37] + q[43];
    }
    if (acc == 12345.678) *sink = acc;
}
__global__ void cal_write8(double* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
__global__ void cal_scatter384(double* p, size_t nrec) {
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += (size_t)gridDim.x * blockDim.x) {
        double* q = p + r * 48;
        q[0] = 1; q[9] = 2; q[17] = 3; q[30] = 4; q[37] = 5; q[43] = 6;
    }
}
int main() {
    const size_t bytes = 512ull << 20;
    double *buf, *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(buf, 0, bytes));
    const size_t nrec = bytes / 384;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(cal_stream16, dim3(4096), dim3(256), 0, 0, (const double2*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(cal_stream8, dim3(4096), dim3(256), 0, 0, buf, bytes / 8, sink);
        hipLaunchKernelGGL(cal_gather384, dim3(4096), dim3(256), 0, 0, buf, nrec, sink);
        hipLaunchKernelGGL(cal_write8, dim3(4096), dim3(256), 0, 0, buf, bytes / 8);
        hipLaunchKernelGGL(cal_scatter384, dim3(4096), dim3(256), 0, 0, buf, nrec);
    }
    CK(hipDeviceSynchronize());
    std::printf("{\"bytes\": %zu, \"records\": %zu, \"gather_useful_bytes\": %zu, \"gather_lines_128\": %zu}\n", bytes, nrec, nrec * 48, nrec * 3 * 128);
    return 0;
}
