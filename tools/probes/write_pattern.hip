// Write-bandwidth probe: n rows of `row_bytes` at stride `row_bytes` (dense [n][row] output), visited in a
// scattered order; each wave writes one piece of `piece` bytes of a row with 16-byte stores.  Pieces of one
// row are either spread over workgroups far apart (split = 1, the way a band-chunked gather writes) or
// written back to back by the same wave (split = 0).  Prints GB/s for each shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(void *dst, v4i v) { asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(dst), "v"(v) : "memory"); }

// the piece [dst, dst + len): dword stores up to the first 16-byte boundary, aligned 16-byte stores, dword stores after the last
__device__ __forceinline__ void write_piece(char *dst, int len, int lane, int aligned, v4i v, char *edges = nullptr, long edge_slot = 0)
{
    if (!aligned) {
        for (int o = lane * 16; o + 16 <= len; o += 1024) store16(dst + o, v);
        return;
    }
    if (aligned >= 2) {
        // whole 64-byte sectors of the piece only; aligned == 3: its two ragged ends go to a 64-byte slot each of a side buffer
        char *lo = (char *)(((size_t)dst + 63) & ~(size_t)63), *hi = (char *)(((size_t)dst + len) & ~(size_t)63);
        for (char *q = lo + lane * 16; q + 16 <= hi; q += 1024) *(v4i *)q = v;
        if (aligned == 3 && lane < 8) *(v4i *)(edges + (size_t)edge_slot * 128 + lane * 16) = v;
        return;
    }
    const int head = (int)((16 - ((size_t)dst & 15)) & 15);
    const int body = (len - head) & ~15;
    if (lane * 4 < head) *(int *)(dst + lane * 4) = v.x;
    for (int o = lane * 16; o + 16 <= body; o += 1024) *(v4i *)(dst + head + o) = v;
    const int tail = len - head - body;
    if (lane * 4 < tail) *(int *)(dst + head + body + lane * 4) = v.x;
}

__global__ __launch_bounds__(256) void scatter_rows(char *out, const int *perm, long n, int row_bytes, int piece, int pieces, int split, int aligned, char *edges)
{
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const v4i v = {lane, 1, 2, 3};
    if (split) {
        // wave -> (piece index major, row minor): the pieces of a row are written by waves far apart
        const long total = n * pieces;
        if (wave >= total) return;
        const int pc = (int)(wave / n);
        const long row = perm[wave % n];
        const int len = min(piece, row_bytes - pc * piece);
        char *dst = out + row * row_bytes + (long)pc * piece;
        write_piece(dst, len, lane, aligned, v, edges, row * pieces + pc);
    } else {
        if (wave >= n) return;
        const long row = perm[wave];
        char *dst = out + row * row_bytes;
        write_piece(dst, row_bytes, lane, aligned, v, edges, row);
    }
}

int main()
{
    const long n = 207400;
    std::vector<int> perm(n);
    for (long i = 0; i < n; i++) perm[i] = (int)i;
    std::mt19937 g(1);
    std::shuffle(perm.begin(), perm.end(), g);
    int *dperm;
    char *out;
    (void)hipMalloc(&dperm, n * 4);
    (void)hipMemcpy(dperm, perm.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, n * 7200 + 4096);
    struct Case { const char *name; int row, piece, split, aligned; } cases[] = {
        {"row 3708, pieces 1008, other workgroups", 3708, 1008, 1},
        {"row 3708, whole row by one wave", 3708, 3708, 0},
        {"row 4096, pieces 1024 (whole lines), other workgroups", 4096, 1024, 1},
        {"row 4096, whole row by one wave", 4096, 4096, 0},
        {"row 3712 (16-byte rows), pieces 1024, other workgroups", 3712, 1024, 1},
        {"row 3708, pieces 2016, other workgroups", 3708, 2016, 1},
        {"row 3708, pieces 1008, other workgroups, aligned stores", 3708, 1008, 1, 1},
        {"row 3708, whole row by one wave, aligned stores", 3708, 3708, 0, 1},
        {"row 3708, pieces 2016, other workgroups, aligned stores", 3708, 2016, 1, 1},
        {"row 3708, whole row by one wave, whole sectors only", 3708, 3708, 0, 2},
        {"row 3708, whole row by one wave, ends to a side buffer", 3708, 3708, 0, 3},
        {"row 3708, pieces 1008, whole sectors only", 3708, 1008, 1, 2},
        {"row 7200 (z=200), pieces 1008, aligned stores", 7200, 1008, 1, 1},
        {"row 7200 (z=200), pieces 1008", 7200, 1008, 1, 0},
    };
    char *edges;
    (void)hipMalloc(&edges, n * 4 * 128);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (auto &c : cases) {
        const int pieces = (c.row + c.piece - 1) / c.piece;
        const long waves = c.split ? n * pieces : n;
        const unsigned blocks = (unsigned)((waves + 3) / 4);
        float best = 1e9f;
        for (int t = 0; t < 6; t++) {
            (void)hipEventRecord(a);
            scatter_rows<<<blocks, 256>>>(out, dperm, n, c.row, c.piece, pieces, c.split, c.aligned, edges);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            if (t) best = std::min(best, ms);
        }
        printf("%-60s %8.1f us  %7.0f GB/s written\n", c.name, best * 1e3, (double)n * c.row / best / 1e6);
    }
    return 0;
}
