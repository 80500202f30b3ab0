// kmu_ingest.hip -- the step before the path: FASTQ text -> (bases, offsets) of the accepted reads.
//
// Reference: needletail's FASTX reader feeds `readblockseq` (src/bin/datasketcher.rs:358-388) and
// `parse_with_needletail` (src/io.rs:12-72): for every record, `count_non_acgt(seq)`; a record with any byte outside
// ACGTacgt is dropped (and counted), the others become `Sequence::new(seq, 2)` in file order.  That stage is serial
// on the host in the reference; here the whole text is processed on the device:
//   1. newline census: every wave counts the '\n' of its 16 KiB region; exclusive scan -> first line number per region
//   2. line walk: the newline that ends line 4r marks the start of read r's bases, the one that ends line 4r + 1
//      its end (a preceding '\r' is dropped); the first bytes of header and separator lines are checked ('@', '+')
//   3. filter: non-ACGT bytes per record (same 16-bytes-per-lane classifier as the kernels of the path)
//   4. scans of the keep flags and kept lengths, then a copy of the kept reads into one dense array.
// Four-line FASTQ records only (what the ONT / Illumina inputs of the reference's tools are); multi-line FASTA is
// not handled.
#include <algorithm>

#include "kmu_ctx.hpp"
#include "kmu_stream.h"

namespace kmu {

static constexpr uint32_t ING_REGION = 16384; // bytes walked by one wave
enum : uint32_t { IERR_HEADER = 1u, IERR_SEPARATOR = 2u };

// 16-bit mask of the bytes of an aligned 16-byte chunk at `pos` that are '\n' (bytes at or past n are ignored)
__device__ __forceinline__ uint32_t newline_mask16(const uint8_t *text, uint64_t pos, uint64_t n) {
    if (pos >= n) return 0u;
    const uint4 v = load_chunk16(text, pos, n); // padded with 'A' past the end
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t t = w[i] ^ 0x0A0A0A0Au;
        const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu); // 0x80 in every zero byte, exactly
        mask |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * i);
    }
    return mask;
}

__global__ void __launch_bounds__(256) k_ing_count(const uint8_t *text, uint64_t n, uint64_t n_regions, uint32_t *cnt) {
    const uint64_t wave_global = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave_global; r < n_regions; r += nwaves) {
        uint32_t c = 0;
        for (uint32_t s = 0; s < ING_REGION; s += 1024)
            c += (uint32_t) __popc(newline_mask16(text, r * ING_REGION + s + 16u * (uint32_t) lane_id(), n));
        c = wave_incl_scan_u32(c);
        if (lane_id() == 63) cnt[r] = c;
    }
}

// exclusive scan of n values by one workgroup; out[n] = total
template <typename T>
__global__ void __launch_bounds__(1024) k_ing_scan(const T *in, uint64_t n, uint64_t *out) {
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = lane_id();
    for (uint64_t base = 0; base < n; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < n ? (uint64_t) in[i] : 0ull;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = ((uint64_t) (uint32_t) __shfl_up((int) (incl >> 32), d, 64) << 32) |
                               (uint32_t) __shfl_up((int) (uint32_t) incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint64_t pre = carry;
        for (int w = 0; w < wave; w++) pre += wsum[w];
        if (i < n) out[i] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

// line walk: seq_start[r] / seq_end[r] of every record
__global__ void __launch_bounds__(256) k_ing_lines(const uint8_t *text, uint64_t n, uint64_t n_regions, const uint64_t *line_base,
                                                   uint64_t n_records, uint64_t *seq_start, uint64_t *seq_end, uint32_t *err) {
    const uint64_t wave_global = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    uint32_t bad = 0;
    for (uint64_t r = wave_global; r < n_regions; r += nwaves) {
        uint64_t line = line_base[r]; // number of the line that the next newline of this region ends (uniform)
        for (uint32_t s = 0; s < ING_REGION; s += 1024) {
            const uint64_t pos0 = r * ING_REGION + s + 16u * (uint32_t) lane_id();
            uint32_t mask = newline_mask16(text, pos0, n);
            const uint32_t c = (uint32_t) __popc(mask);
            const uint32_t incl = wave_incl_scan_u32(c);
            uint64_t l = line + incl - c;
            line += (uint64_t) bcast_u32(incl, 63);
            while (mask) {
                const uint32_t j = (uint32_t) __ffs((int) mask) - 1u;
                mask &= mask - 1u;
                const uint64_t p = pos0 + j; // this newline ends line l; line l + 1 starts at p + 1
                const uint64_t rec = l >> 2;
                const uint32_t kind = (uint32_t) l & 3u;
                if (rec < n_records) {
                    if (kind == 0u) seq_start[rec] = p + 1;
                    if (kind == 1u) seq_end[rec] = (p > 0 && text[p - 1] == '\r') ? p - 1 : p;
                }
                if (p + 1 < n) {
                    const uint64_t nrec = (l + 1) >> 2;
                    const uint32_t nkind = (uint32_t) (l + 1) & 3u;
                    if (nrec < n_records) {
                        if (nkind == 0u && text[p + 1] != '@') bad |= IERR_HEADER;
                        if (nkind == 2u && text[p + 1] != '+') bad |= IERR_SEPARATOR;
                    }
                }
                l++;
            }
        }
    }
    if (wave_global == 0 && lane_id() == 0 && n_records && text[0] != '@') bad |= IERR_HEADER;
    if (bad) atomicOr(err, bad);
}

// per record: non-ACGT bytes; keep flag and kept length; totals[0] += bases, [1] += bad bases, [2] += bad reads
__global__ void __launch_bounds__(256) k_ing_filter(const uint8_t *text, uint64_t n, const uint64_t *seq_start,
                                                    const uint64_t *seq_end, uint64_t n_records, uint32_t *keep,
                                                    uint64_t *kept_len, unsigned long long *totals) {
    __shared__ uint32_t acc;
    for (uint64_t r = blockIdx.x; r < n_records; r += gridDim.x) {
        SeqView s;
        s.base = text;
        s.begin = seq_start[r];
        s.len = seq_end[r] - seq_start[r];
        s.total = n;
        s.packed = 0;
        if (threadIdx.x == 0) acc = 0;
        __syncthreads();
        const uint64_t nwords = seq_num_words(s);
        uint32_t local = 0;
        for (uint64_t w = threadIdx.x; w < nwords; w += blockDim.x) {
            uint32_t b;
            (void) load_code_word(s, w, b);
            local += (uint32_t) __popc(b);
        }
        if (local) atomicAdd(&acc, local);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t nb_bad = acc;
            keep[r] = nb_bad == 0u;
            kept_len[r] = nb_bad == 0u ? s.len : 0ull;
            atomicAdd(&totals[0], (unsigned long long) s.len);
            if (nb_bad) {
                atomicAdd(&totals[1], (unsigned long long) nb_bad);
                atomicAdd(&totals[2], 1ull);
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_ing_copy(const uint8_t *text, const uint64_t *seq_start, const uint32_t *keep,
                                                  const uint64_t *kept_len, const uint64_t *rank, const uint64_t *out_off,
                                                  uint64_t n_records, uint8_t *bases_out, uint64_t *offsets_out,
                                                  uint32_t *record_index_out) {
    for (uint64_t r = blockIdx.x; r < n_records; r += gridDim.x) {
        if (!keep[r]) continue;
        const uint64_t src = seq_start[r], dst = out_off[r], len = kept_len[r];
        for (uint64_t i = threadIdx.x; i < len; i += blockDim.x) bases_out[dst + i] = text[src + i];
        if (threadIdx.x == 0) {
            offsets_out[rank[r]] = dst;
            if (record_index_out) record_index_out[rank[r]] = (uint32_t) r;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets_out[rank[n_records]] = out_off[n_records];
}

} // namespace kmu

using namespace kmu;

extern "C" int kmu_ingest_fastq(kmu_ctx *ctx, const uint8_t *text, uint64_t n_bytes, int mem, uint8_t *bases_out,
                                uint64_t bases_cap, uint64_t *offsets_out, uint64_t offsets_cap, uint32_t *record_index_out,
                                kmu_ingest_info *info) {
    if (!ctx || !info || (n_bytes && !text)) return fail(ctx, KMU_E_BAD_ARG, "null argument");
    memset(info, 0, sizeof *info);
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (n_bytes == 0) {
        if (offsets_out && offsets_cap >= 1) {
            const uint64_t zero = 0;
            if (mem == KMU_MEM_HOST) offsets_out[0] = 0;
            else KMU_HIP(ctx, hipMemcpyAsync(offsets_out, &zero, 8, hipMemcpyHostToDevice, ctx->stream));
            KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        return KMU_OK;
    }
    const uint8_t *d_text = text;
    if (mem == KMU_MEM_DEVICE && ((uintptr_t) text & 15u) != 0) return fail(ctx, KMU_E_BAD_ARG, "device `text` must be 16-byte aligned");
    if (mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "ing.text", n_bytes + 64, &q));
        KMU_HIP(ctx, hipMemcpyAsync(q, text, n_bytes, hipMemcpyHostToDevice, ctx->stream));
        d_text = (const uint8_t *) q;
    }
    const uint64_t n_regions = (n_bytes + ING_REGION - 1) / ING_REGION;
    void *cnt, *lbase, *scal;
    KMU_TRY(dev_buf(ctx, "ing.cnt", n_regions * 4, &cnt));
    KMU_TRY(dev_buf(ctx, "ing.lbase", (n_regions + 1) * 8, &lbase));
    KMU_TRY(dev_buf(ctx, "ing.scal", 64, &scal)); // [0..2] totals, [4] error word
    KMU_HIP(ctx, hipMemsetAsync(scal, 0, 64, ctx->stream));
    const int grid_w = (int) std::min<uint64_t>((n_regions + 3) / 4, (uint64_t) ctx->num_cus * 8);
    {
        KernelTimer t(ctx, "k_ing_count");
        hipLaunchKernelGGL(k_ing_count, dim3(grid_w), dim3(256), 0, ctx->stream, d_text, n_bytes, n_regions, (uint32_t *) cnt);
    }
    {
        KernelTimer t(ctx, "k_ing_scan");
        hipLaunchKernelGGL(k_ing_scan<uint32_t>, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *) cnt, n_regions,
                           (uint64_t *) lbase);
    }
    uint64_t total_nl = 0;
    uint8_t last = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&total_nl, (const uint64_t *) lbase + n_regions, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipMemcpyAsync(&last, d_text + n_bytes - 1, 1, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t n_lines = total_nl + (last != '\n' ? 1 : 0);
    if (n_lines % 4 != 0)
        return fail(ctx, KMU_E_BAD_ARG, "invalid record: %llu lines are not a whole number of 4-line FASTQ records",
                    (unsigned long long) n_lines);
    const uint64_t n_records = n_lines / 4;
    if (n_records > 0xFFFFFFFFull) return fail(ctx, KMU_E_UNSUPPORTED, "more than 2^32 records in one call");
    info->n_records = n_records;
    void *sstart, *send, *keep, *klen, *rank, *ooff;
    KMU_TRY(dev_buf(ctx, "ing.sstart", (n_records + 1) * 8, &sstart));
    KMU_TRY(dev_buf(ctx, "ing.send", (n_records + 1) * 8, &send));
    KMU_TRY(dev_buf(ctx, "ing.keep", (n_records + 1) * 4, &keep));
    KMU_TRY(dev_buf(ctx, "ing.klen", (n_records + 1) * 8, &klen));
    KMU_TRY(dev_buf(ctx, "ing.rank", (n_records + 1) * 8, &rank));
    KMU_TRY(dev_buf(ctx, "ing.ooff", (n_records + 1) * 8, &ooff));
    uint32_t *d_err = (uint32_t *) scal + 8;
    {
        KernelTimer t(ctx, "k_ing_lines");
        hipLaunchKernelGGL(k_ing_lines, dim3(grid_w), dim3(256), 0, ctx->stream, d_text, n_bytes, n_regions,
                           (const uint64_t *) lbase, n_records, (uint64_t *) sstart, (uint64_t *) send, d_err);
    }
    const int grid_r = (int) std::min<uint64_t>(std::max<uint64_t>(n_records, 1), (uint64_t) ctx->num_cus * 16);
    {
        KernelTimer t(ctx, "k_ing_filter");
        hipLaunchKernelGGL(k_ing_filter, dim3(grid_r), dim3(256), 0, ctx->stream, d_text, n_bytes, (const uint64_t *) sstart,
                           (const uint64_t *) send, n_records, (uint32_t *) keep, (uint64_t *) klen,
                           (unsigned long long *) scal);
    }
    {
        KernelTimer t(ctx, "k_ing_scan");
        hipLaunchKernelGGL(k_ing_scan<uint32_t>, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *) keep, n_records,
                           (uint64_t *) rank);
        hipLaunchKernelGGL(k_ing_scan<uint64_t>, dim3(1), dim3(1024), 0, ctx->stream, (const uint64_t *) klen, n_records,
                           (uint64_t *) ooff);
    }
    KMU_HIP(ctx, hipGetLastError());
    uint64_t h_scal[8], n_kept = 0, kept_bases = 0;
    KMU_HIP(ctx, hipMemcpyAsync(h_scal, scal, 64, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipMemcpyAsync(&n_kept, (const uint64_t *) rank + n_records, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipMemcpyAsync(&kept_bases, (const uint64_t *) ooff + n_records, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t errw = (uint32_t) h_scal[4];
    if (errw & IERR_HEADER) return fail(ctx, KMU_E_BAD_ARG, "invalid record: a header line does not start with '@'");
    if (errw & IERR_SEPARATOR) return fail(ctx, KMU_E_BAD_ARG, "invalid record: a separator line does not start with '+'");
    info->n_kept = n_kept;
    info->kept_bases = kept_bases;
    info->n_bases = h_scal[0];
    info->nb_bad_bases = h_scal[1];
    info->nb_bad_reads = h_scal[2];
    if (!bases_out && !offsets_out) return KMU_OK; // sizes only
    if (!bases_out || !offsets_out) return fail(ctx, KMU_E_BAD_ARG, "bases_out and offsets_out go together");
    if (bases_cap < kept_bases || offsets_cap < n_kept + 1)
        return fail(ctx, KMU_E_BAD_ARG, "output too small: need %llu bases and %llu offsets", (unsigned long long) kept_bases,
                    (unsigned long long) (n_kept + 1));
    uint8_t *d_bases = bases_out;
    uint64_t *d_off = offsets_out;
    uint32_t *d_idx = record_index_out;
    if (mem == KMU_MEM_HOST) {
        void *q;
        KMU_TRY(dev_buf(ctx, "ing.bases", kept_bases + 64, &q));
        d_bases = (uint8_t *) q;
        KMU_TRY(dev_buf(ctx, "ing.off", (n_kept + 1) * 8, &q));
        d_off = (uint64_t *) q;
        if (record_index_out) {
            KMU_TRY(dev_buf(ctx, "ing.idx", (n_kept + 1) * 4, &q));
            d_idx = (uint32_t *) q;
        }
    }
    {
        KernelTimer t(ctx, "k_ing_copy");
        hipLaunchKernelGGL(k_ing_copy, dim3(grid_r), dim3(256), 0, ctx->stream, d_text, (const uint64_t *) sstart,
                           (const uint32_t *) keep, (const uint64_t *) klen, (const uint64_t *) rank, (const uint64_t *) ooff,
                           n_records, d_bases, d_off, d_idx);
    }
    KMU_HIP(ctx, hipGetLastError());
    if (mem == KMU_MEM_HOST) {
        KMU_HIP(ctx, hipMemcpyAsync(bases_out, d_bases, kept_bases, hipMemcpyDeviceToHost, ctx->stream));
        KMU_HIP(ctx, hipMemcpyAsync(offsets_out, d_off, (n_kept + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (record_index_out)
            KMU_HIP(ctx, hipMemcpyAsync(record_index_out, d_idx, n_kept * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    return finish_call(ctx, mem);
}
