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
// FASTQ: four-line records (what needletail accepts).  FASTA (kmu_ingest_fasta; needletail's other format, what gsearch
// feeds): a line that starts with '>' opens a record, the lines up to the next such line are its sequence with the line
// ends ("\n", "\r\n") removed -- record.seq() of needletail.  The device version numbers the lines as above, marks
// header lines, scans header flags and sequence-line lengths, joins the sequence lines into one dense stream (every
// region of the text walked by one wave, whatever the line lengths: 60-column genomes and one-line chromosomes alike),
// and hands the stream with its record bounds to steps 3 and 4.  kmu_ingest_fastx picks the format from the first byte
// like needletail::parse_fastx_file.
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

// The same scan for long arrays (the lines of a genome-sized FASTA text): tiles of 8192 values, tile sums, the
// single-workgroup scan above over the sums, then every tile scans itself from its offset.
static constexpr uint32_t SCAN_ITEMS = 8, SCAN_TILE = 1024 * SCAN_ITEMS;

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = ((uint64_t) (uint32_t) __shfl_up((int) (v >> 32), d, 64) << 32) | (uint32_t) __shfl_up((int) (uint32_t) v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

template <typename T>
__global__ void __launch_bounds__(1024) k_scan_tile_sums(const T *in, uint64_t n, uint64_t *tile_sum) {
    __shared__ uint64_t wsum[16];
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_ITEMS; j++) s += base + j < n ? (uint64_t) in[base + j] : 0ull;
    s = wave_incl_scan_u64(s);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < 16; w++) t += wsum[w];
        tile_sum[blockIdx.x] = t;
    }
}

template <typename T>
__global__ void __launch_bounds__(1024) k_scan_tiles(const T *in, uint64_t n, const uint64_t *tile_off, uint64_t n_tiles,
                                                     uint64_t *out) {
    __shared__ uint64_t wsum[16];
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_ITEMS; j++) {
        v[j] = base + j < n ? (uint64_t) in[base + j] : 0ull;
        s += v[j];
    }
    const uint64_t incl = wave_incl_scan_u64(s);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t run = tile_off[blockIdx.x] + incl - s;
    for (int w = 0; w < (int) (threadIdx.x >> 6); w++) run += wsum[w];
#pragma unroll
    for (uint32_t j = 0; j < SCAN_ITEMS; j++) {
        if (base + j < n) out[base + j] = run;
        run += v[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tile_off[n_tiles];
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

// ---- FASTA ---------------------------------------------------------------------------------------------------------------
enum : uint32_t { IERR_FASTA_START = 4u };

// line l occupies text[line_start[l], line_end[l]) (line end and a preceding '\r' excluded); is_hdr[l] = starts with '>'
__global__ void __launch_bounds__(256) k_fa_lines(const uint8_t *text, uint64_t n, uint64_t n_regions, const uint64_t *line_base,
                                                  uint64_t n_lines, uint64_t *line_start, uint64_t *line_end, uint32_t *is_hdr) {
    const uint64_t wave_global = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave_global; r < n_regions; r += nwaves) {
        uint64_t line = line_base[r];
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
                const uint64_t p = pos0 + j; // this newline ends line l
                line_end[l] = (p > 0 && text[p - 1] == '\r') ? p - 1 : p;
                if (l + 1 < n_lines) {
                    line_start[l + 1] = p + 1;
                    is_hdr[l + 1] = text[p + 1] == '>';
                }
                l++;
            }
        }
    }
    if (wave_global == 0 && lane_id() == 0) {
        line_start[0] = 0;
        is_hdr[0] = text[0] == '>';
        if (text[n - 1] != '\n') line_end[n_lines - 1] = n; // last line without a line end
    }
}

// bytes a line contributes to the sequence stream
__global__ void __launch_bounds__(256) k_fa_lens(const uint64_t *line_start, const uint64_t *line_end, const uint32_t *is_hdr,
                                                 uint64_t n_lines, uint64_t *seqlen) {
    for (uint64_t l = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; l < n_lines; l += (uint64_t) gridDim.x * blockDim.x)
        seqlen[l] = is_hdr[l] ? 0ull : line_end[l] - line_start[l];
}

// record r (the r-th header line) owns stream[rec_start[r], rec_end[r])
__global__ void __launch_bounds__(256) k_fa_records(const uint32_t *is_hdr, const uint64_t *hdr_rank, const uint64_t *seqpos,
                                                    uint64_t n_lines, uint64_t n_records, uint64_t *rec_start, uint64_t *rec_end) {
    for (uint64_t l = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; l < n_lines; l += (uint64_t) gridDim.x * blockDim.x) {
        if (!is_hdr[l]) continue;
        const uint64_t r = hdr_rank[l];
        rec_start[r] = seqpos[l];
        if (r > 0) rec_end[r - 1] = seqpos[l];
        if (r + 1 == n_records) rec_end[r] = seqpos[n_lines];
    }
}

// the sequence lines, joined: byte i of the text, on sequence line l, goes to stream[seqpos[l] + i - line_start[l]]
__global__ void __launch_bounds__(256) k_fa_join(const uint8_t *text, uint64_t n, uint64_t n_regions, const uint64_t *line_base,
                                                 const uint64_t *line_start, const uint64_t *line_end, const uint32_t *is_hdr,
                                                 const uint64_t *seqpos, uint8_t *stream) {
    const uint64_t wave_global = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave_global; r < n_regions; r += nwaves) {
        uint64_t line = line_base[r];
        for (uint32_t s = 0; s < ING_REGION; s += 1024) {
            const uint64_t pos0 = r * ING_REGION + s + 16u * (uint32_t) lane_id();
            const uint32_t mask = newline_mask16(text, pos0, n);
            const uint32_t c = (uint32_t) __popc(mask);
            const uint32_t incl = wave_incl_scan_u32(c);
            uint64_t l = line + incl - c; // line of this lane's first byte
            line += (uint64_t) bcast_u32(incl, 63);
            if (pos0 >= n) continue;
            const uint4 v = load_chunk16(text, pos0, n);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint64_t ls = line_start[l], le = line_end[l], sp = seqpos[l];
            bool hdr = is_hdr[l] != 0u;
            for (int j = 0; j < 16 && pos0 + j < n; j++) {
                const uint64_t i = pos0 + j;
                if (!hdr && i < le) stream[sp + (i - ls)] = (uint8_t) (w[j >> 2] >> (8 * (j & 3)));
                if ((mask >> j) & 1u) { // the next byte opens line l + 1
                    l++;
                    if (i + 1 < n) {
                        ls = line_start[l]; le = line_end[l]; sp = seqpos[l];
                        hdr = is_hdr[l] != 0u;
                    }
                }
            }
        }
    }
}

} // namespace kmu

using namespace kmu;

// exclusive scan of n values on the context's stream; out[n] = total
template <typename T> static int device_scan(kmu_ctx *ctx, const T *in, uint64_t n, uint64_t *out) {
    KernelTimer t(ctx, "k_ing_scan");
    if (n <= 4 * SCAN_TILE) {
        hipLaunchKernelGGL(k_ing_scan<T>, dim3(1), dim3(1024), 0, ctx->stream, in, n, out);
        return KMU_OK;
    }
    const uint64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    void *tsum, *toff;
    KMU_TRY(dev_buf(ctx, "ing.tsum", (n_tiles + 1) * 8, &tsum));
    KMU_TRY(dev_buf(ctx, "ing.toff", (n_tiles + 1) * 8, &toff));
    hipLaunchKernelGGL(k_scan_tile_sums<T>, dim3((uint32_t) n_tiles), dim3(1024), 0, ctx->stream, in, n, (uint64_t *) tsum);
    hipLaunchKernelGGL(k_ing_scan<uint64_t>, dim3(1), dim3(1024), 0, ctx->stream, (const uint64_t *) tsum, n_tiles, (uint64_t *) toff);
    hipLaunchKernelGGL(k_scan_tiles<T>, dim3((uint32_t) n_tiles), dim3(1024), 0, ctx->stream, in, n, (const uint64_t *) toff, n_tiles, out);
    return KMU_OK;
}

namespace kmu {
// exclusive scan of n u32 values into u64 (out[n] = total), for the other modules
int device_scan_u32(kmu_ctx *ctx, const uint32_t *in, uint64_t n, uint64_t *out) { return device_scan<uint32_t>(ctx, in, n, out); }
}

// steps 3 and 4 shared by both formats: filter the records [seq_start, seq_end) of `d_text`, scan, copy out
static int ingest_filter_copy(kmu_ctx *ctx, const uint8_t *d_text, uint64_t n_text, uint64_t n_records, void *sstart, void *send,
                              void *scal, int mem, uint8_t *bases_out, uint64_t bases_cap, uint64_t *offsets_out,
                              uint64_t offsets_cap, uint32_t *record_index_out, kmu_ingest_info *info) {
    void *keep, *klen, *rank, *ooff;
    KMU_TRY(dev_buf(ctx, "ing.keep", (n_records + 1) * 4, &keep));
    KMU_TRY(dev_buf(ctx, "ing.klen", (n_records + 1) * 8, &klen));
    KMU_TRY(dev_buf(ctx, "ing.rank", (n_records + 1) * 8, &rank));
    KMU_TRY(dev_buf(ctx, "ing.ooff", (n_records + 1) * 8, &ooff));
    const int grid_r = (int) std::min<uint64_t>(std::max<uint64_t>(n_records, 1), (uint64_t) ctx->num_cus * 16);
    {
        KernelTimer t(ctx, "k_ing_filter");
        hipLaunchKernelGGL(k_ing_filter, dim3(grid_r), dim3(256), 0, ctx->stream, d_text, n_text, (const uint64_t *) sstart,
                           (const uint64_t *) send, n_records, (uint32_t *) keep, (uint64_t *) klen,
                           (unsigned long long *) scal);
    }
    KMU_TRY(device_scan(ctx, (const uint32_t *) keep, n_records, (uint64_t *) rank));
    KMU_TRY(device_scan(ctx, (const uint64_t *) klen, n_records, (uint64_t *) ooff));
    KMU_HIP(ctx, hipGetLastError());
    uint64_t h_scal[8], n_kept = 0, kept_bases = 0;
    KMU_HIP(ctx, hipMemcpyAsync(h_scal, scal, 64, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipMemcpyAsync(&n_kept, (const uint64_t *) rank + n_records, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipMemcpyAsync(&kept_bases, (const uint64_t *) ooff + n_records, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t errw = (uint32_t) h_scal[4];
    if (errw & IERR_HEADER) return fail(ctx, KMU_E_BAD_ARG, "invalid record: a header line does not start with '@'");
    if (errw & IERR_SEPARATOR) return fail(ctx, KMU_E_BAD_ARG, "invalid record: a separator line does not start with '+'");
    info->n_records = n_records;
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

// text to the device (host mode), newline census and its scan; *n_lines_out = lines of the text
static int ingest_lines(kmu_ctx *ctx, const uint8_t *text, uint64_t n_bytes, int mem, const uint8_t **d_text_out, void **lbase_out,
                        void **scal_out, uint64_t *n_regions_out, int *grid_out, uint64_t *n_lines_out) {
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
    KMU_TRY(device_scan(ctx, (const uint32_t *) cnt, n_regions, (uint64_t *) lbase));
    uint64_t total_nl = 0;
    uint8_t last = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&total_nl, (const uint64_t *) lbase + n_regions, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipMemcpyAsync(&last, d_text + n_bytes - 1, 1, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *d_text_out = d_text;
    *lbase_out = lbase;
    *scal_out = scal;
    *n_regions_out = n_regions;
    *grid_out = grid_w;
    *n_lines_out = total_nl + (last != '\n' ? 1 : 0);
    return KMU_OK;
}

static int ingest_empty(kmu_ctx *ctx, int mem, uint64_t *offsets_out, uint64_t offsets_cap) {
    if (offsets_out && offsets_cap >= 1) {
        const uint64_t zero = 0;
        if (mem == KMU_MEM_HOST) offsets_out[0] = 0;
        else KMU_HIP(ctx, hipMemcpyAsync(offsets_out, &zero, 8, hipMemcpyHostToDevice, ctx->stream));
        KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return KMU_OK;
}

extern "C" int kmu_ingest_fastq(kmu_ctx *ctx, const uint8_t *text, uint64_t n_bytes, int mem, uint8_t *bases_out,
                                uint64_t bases_cap, uint64_t *offsets_out, uint64_t offsets_cap, uint32_t *record_index_out,
                                kmu_ingest_info *info) {
    if (!ctx || !info || (n_bytes && !text)) return fail(ctx, KMU_E_BAD_ARG, "null argument");
    memset(info, 0, sizeof *info);
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (n_bytes == 0) return ingest_empty(ctx, mem, offsets_out, offsets_cap);
    const uint8_t *d_text;
    void *lbase, *scal;
    uint64_t n_regions, n_lines;
    int grid_w;
    KMU_TRY(ingest_lines(ctx, text, n_bytes, mem, &d_text, &lbase, &scal, &n_regions, &grid_w, &n_lines));
    if (n_lines % 4 != 0)
        return fail(ctx, KMU_E_BAD_ARG, "invalid record: %llu lines are not a whole number of 4-line FASTQ records",
                    (unsigned long long) n_lines);
    const uint64_t n_records = n_lines / 4;
    if (n_records > 0xFFFFFFFFull) return fail(ctx, KMU_E_UNSUPPORTED, "more than 2^32 records in one call");
    void *sstart, *send;
    KMU_TRY(dev_buf(ctx, "ing.sstart", (n_records + 1) * 8, &sstart));
    KMU_TRY(dev_buf(ctx, "ing.send", (n_records + 1) * 8, &send));
    {
        KernelTimer t(ctx, "k_ing_lines");
        hipLaunchKernelGGL(k_ing_lines, dim3(grid_w), dim3(256), 0, ctx->stream, d_text, n_bytes, n_regions,
                           (const uint64_t *) lbase, n_records, (uint64_t *) sstart, (uint64_t *) send, (uint32_t *) scal + 8);
    }
    return ingest_filter_copy(ctx, d_text, n_bytes, n_records, sstart, send, scal, mem, bases_out, bases_cap, offsets_out,
                              offsets_cap, record_index_out, info);
}

extern "C" int kmu_ingest_fasta(kmu_ctx *ctx, const uint8_t *text, uint64_t n_bytes, int mem, uint8_t *bases_out,
                                uint64_t bases_cap, uint64_t *offsets_out, uint64_t offsets_cap, uint32_t *record_index_out,
                                kmu_ingest_info *info) {
    if (!ctx || !info || (n_bytes && !text)) return fail(ctx, KMU_E_BAD_ARG, "null argument");
    memset(info, 0, sizeof *info);
    KMU_HIP(ctx, hipSetDevice(ctx->device));
    if (n_bytes == 0) return ingest_empty(ctx, mem, offsets_out, offsets_cap);
    const uint8_t *d_text;
    void *lbase, *scal;
    uint64_t n_regions, n_lines;
    int grid_w;
    KMU_TRY(ingest_lines(ctx, text, n_bytes, mem, &d_text, &lbase, &scal, &n_regions, &grid_w, &n_lines));
    void *lstart, *lend, *hdr, *slen, *hrank, *spos;
    KMU_TRY(dev_buf(ctx, "ing.lstart", (n_lines + 1) * 8, &lstart));
    KMU_TRY(dev_buf(ctx, "ing.lend", (n_lines + 1) * 8, &lend));
    KMU_TRY(dev_buf(ctx, "ing.hdr", (n_lines + 1) * 4, &hdr));
    KMU_TRY(dev_buf(ctx, "ing.slen", (n_lines + 1) * 8, &slen));
    KMU_TRY(dev_buf(ctx, "ing.hrank", (n_lines + 1) * 8, &hrank));
    KMU_TRY(dev_buf(ctx, "ing.spos", (n_lines + 1) * 8, &spos));
    const int grid_l = (int) std::min<uint64_t>((n_lines + 255) / 256, (uint64_t) ctx->num_cus * 16);
    {
        KernelTimer t(ctx, "k_fa_lines");
        hipLaunchKernelGGL(k_fa_lines, dim3(grid_w), dim3(256), 0, ctx->stream, d_text, n_bytes, n_regions, (const uint64_t *) lbase,
                           n_lines, (uint64_t *) lstart, (uint64_t *) lend, (uint32_t *) hdr);
        hipLaunchKernelGGL(k_fa_lens, dim3(grid_l), dim3(256), 0, ctx->stream, (const uint64_t *) lstart, (const uint64_t *) lend,
                           (const uint32_t *) hdr, n_lines, (uint64_t *) slen);
    }
    KMU_TRY(device_scan(ctx, (const uint32_t *) hdr, n_lines, (uint64_t *) hrank));
    KMU_TRY(device_scan(ctx, (const uint64_t *) slen, n_lines, (uint64_t *) spos));
    uint64_t n_records = 0, n_stream = 0;
    uint32_t first_hdr = 0;
    KMU_HIP(ctx, hipMemcpyAsync(&n_records, (const uint64_t *) hrank + n_lines, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipMemcpyAsync(&n_stream, (const uint64_t *) spos + n_lines, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipMemcpyAsync(&first_hdr, hdr, 4, hipMemcpyDeviceToHost, ctx->stream));
    KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!first_hdr) return fail(ctx, KMU_E_BAD_ARG, "invalid record: a FASTA text starts with '>'");
    if (n_records > 0xFFFFFFFFull) return fail(ctx, KMU_E_UNSUPPORTED, "more than 2^32 records in one call");
    void *sstart, *send, *stream;
    KMU_TRY(dev_buf(ctx, "ing.sstart", (n_records + 1) * 8, &sstart));
    KMU_TRY(dev_buf(ctx, "ing.send", (n_records + 1) * 8, &send));
    KMU_TRY(dev_buf(ctx, "ing.stream", n_stream + 64, &stream));
    {
        KernelTimer t(ctx, "k_fa_join");
        hipLaunchKernelGGL(k_fa_records, dim3(grid_l), dim3(256), 0, ctx->stream, (const uint32_t *) hdr, (const uint64_t *) hrank,
                           (const uint64_t *) spos, n_lines, n_records, (uint64_t *) sstart, (uint64_t *) send);
        hipLaunchKernelGGL(k_fa_join, dim3(grid_w), dim3(256), 0, ctx->stream, d_text, n_bytes, n_regions, (const uint64_t *) lbase,
                           (const uint64_t *) lstart, (const uint64_t *) lend, (const uint32_t *) hdr, (const uint64_t *) spos,
                           (uint8_t *) stream);
    }
    return ingest_filter_copy(ctx, (const uint8_t *) stream, n_stream, n_records, sstart, send, scal, mem, bases_out, bases_cap,
                              offsets_out, offsets_cap, record_index_out, info);
}

// needletail::parse_fastx_file looks at the first byte: '>' FASTA, '@' FASTQ
extern "C" int kmu_ingest_fastx(kmu_ctx *ctx, const uint8_t *text, uint64_t n_bytes, int mem, uint8_t *bases_out,
                                uint64_t bases_cap, uint64_t *offsets_out, uint64_t offsets_cap, uint32_t *record_index_out,
                                kmu_ingest_info *info) {
    if (!ctx || !info || (n_bytes && !text)) return fail(ctx, KMU_E_BAD_ARG, "null argument");
    uint8_t first = '@';
    if (n_bytes) {
        if (mem == KMU_MEM_HOST) first = text[0];
        else {
            KMU_HIP(ctx, hipSetDevice(ctx->device));
            KMU_HIP(ctx, hipMemcpyAsync(&first, text, 1, hipMemcpyDeviceToHost, ctx->stream));
            KMU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    if (first == '>') return kmu_ingest_fasta(ctx, text, n_bytes, mem, bases_out, bases_cap, offsets_out, offsets_cap, record_index_out, info);
    if (first == '@') return kmu_ingest_fastq(ctx, text, n_bytes, mem, bases_out, bases_cap, offsets_out, offsets_cap, record_index_out, info);
    memset(info, 0, sizeof *info);
    return fail(ctx, KMU_E_BAD_ARG, "invalid record: the text starts neither with '>' (FASTA) nor with '@' (FASTQ)");
}
