// lds_dma.h — global -> LDS copies that pass through no register (`global_load_lds_dword`, gfx950), for the speculative
// accumulate kernels (accumulate_quad.hip, accumulate_mid.hip).
//
// One wave-instruction moves 64 x 4 bytes = 32 rows of ONE f64 column: the source is a scalar base (the column's pointer at
// the block's first row) plus the lane's constant byte offset 4 * lane, the destination M0 + 4 * lane — no vector instruction
// computes an address.  The statements are inline asm because M0 must be written in the statement that uses it; the compiler
// therefore neither counts these loads nor waits for them: the caller waits (`s_waitcnt vmcnt(N)` / `lds_dma_wait_all`) before it
// reads what they wrote, and keeps its own ordinary global loads out of the span in which they are outstanding.
#pragma once
#include <hip/hip_runtime.h>

namespace anofox {

// the LDS byte address of a pointer into a wave's slice, as a scalar
__device__ __forceinline__ unsigned lds_dma_address(const double *slice) {
	return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)const_cast<double *>(slice));
}

// one column: 256 bytes from src + voff (per lane) to LDS address dst + 4 * lane.  X4 (`global_load_lds_dwordx4`, new on gfx950):
// 1024 bytes = 128 rows of the column per wave-instruction, 16 bytes per lane to dst + 16 * lane (tools/lds_dma_x4_probe.hip) —
// a request of 1 KB per column where the dword form makes four of 256 bytes.
template <bool X4 = false>
__device__ __forceinline__ void lds_dma1(unsigned voff, unsigned dst, const double *src) {
	unsigned keep;
	if constexpr (X4) {
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
		             : "=&s"(keep)
		             : "v"(voff), "s"(dst), "s"(src)
		             : "memory");
	} else {
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
		             : "=&s"(keep)
		             : "v"(voff), "s"(dst), "s"(src)
		             : "memory");
	}
}

// four columns whose destinations are STRIDE bytes apart: M0 is saved once, stepped per load and restored
template <int STRIDE, bool X4 = false>
__device__ __forceinline__ void lds_dma4(unsigned voff, unsigned dst, const double *c0, const double *c1, const double *c2, const double *c3) {
	unsigned keep;
	if constexpr (X4) {
		asm volatile("s_mov_b32 %0, m0\n\t"
		             "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
		             "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
		             "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
		             "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %6\n\t"
		             "s_mov_b32 m0, %0"
		             : "=&s"(keep)
		             : "v"(voff), "s"(dst), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "n"(STRIDE)
		             : "memory", "scc");
		return;
	}
	asm volatile("s_mov_b32 %0, m0\n\t"
	             "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\t"
	             "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dword %1, %4\n\t"
	             "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dword %1, %5\n\t"
	             "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dword %1, %6\n\t"
	             "s_mov_b32 m0, %0"
	             : "=&s"(keep)
	             : "v"(voff), "s"(dst), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "n"(STRIDE)
	             : "memory", "scc");
}

// the lanes' source offsets for a block of which only `left` (1 .. 31) rows exist: lanes past the last row re-read its two
// dwords, so that nothing beyond the group is touched (the rows past the end are masked by the caller's arithmetic)
__device__ __forceinline__ unsigned lds_dma_offsets(int lane, int64_t left) {
	unsigned voff = (unsigned)lane * 4u;
	if (left < 32) {
		const unsigned last = (unsigned)left * 8u - 4u; // the last row's high dword
		voff = voff < last ? voff : (last - 4u + (voff & 4u));
	}
	return voff;
}

// ---- a whole 32-row block: columns 0 .. ncol - 1 of the kernel's column table -------------------------------------------
// The column pointers live in the kernel-argument segment (WideArgs::x_table, with y stored behind the last feature by the
// host).  Kept resident they need two SGPRs each — 35 .. 65 columns do not fit, the compiler spills them into VGPR lanes and
// every use becomes a v_readlane (the first version of the tile kernel issued 16 vector and 14 scalar instructions per ROW
// that way and ran slower than the kernel it was to replace: profiles/r04_tile_dma_pmc.md).  Here the table is re-read from
// the argument segment for every block, eight pointers per scalar load, one group ahead of the group whose loads are being
// issued; the empty asm statements pin that order (a load of the table cannot move across them).
typedef const double *const __attribute__((address_space(4))) *lds_dma_table_t;

__device__ __forceinline__ lds_dma_table_t lds_dma_table(unsigned byte_offset_in_kernarg) {
	return (lds_dma_table_t)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + byte_offset_in_kernarg);
}

struct LdsDmaPtr8 {
	const double *c[8];
};

__device__ __forceinline__ LdsDmaPtr8 lds_dma_load8(lds_dma_table_t &tab, int c0) {
	unsigned long long t = (unsigned long long)tab;
	asm volatile("" : "+s"(t)); // (the loads below depend on this statement: they stay behind it)
	tab = (lds_dma_table_t)t;
	LdsDmaPtr8 r;
#pragma unroll
	for (int i = 0; i < 8; ++i) r.c[i] = tab[c0 + i];
	return r;
}

// MAXCOLS: compile-time bound of ncol (the loop is unrolled over groups of eight; a group beyond ncol costs one scalar branch)
template <int STRIDE, int MAXCOLS, bool X4 = false>
__device__ __forceinline__ void lds_dma_block(lds_dma_table_t tab, int ncol, int64_t blk, unsigned voff, unsigned dst) {
	constexpr int NG = (MAXCOLS + 7) / 8;
	// (the column count is loop-invariant for the caller's row loop: left alone, the compiler evaluates the ~3 comparisons per
	// group once, keeps the 50 results in scalar register pairs, spills them to vector lanes and reads them back with
	// v_readlane in every block — recomputing a comparison is one scalar instruction)
	asm volatile("" : "+s"(ncol));
	LdsDmaPtr8 cur = lds_dma_load8(tab, 0);
#pragma unroll
	for (int g = 0; g < NG; ++g) {
		if (8 * g >= ncol) break; // wave-uniform
		LdsDmaPtr8 nxt = cur;
		if (g + 1 < NG) nxt = lds_dma_load8(tab, 8 * (g + 1));
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int c = 8 * g + 4 * h;
			const unsigned d = dst + (unsigned)c * (unsigned)STRIDE;
			if (c + 4 <= ncol) {
				lds_dma4<STRIDE, X4>(voff, d, cur.c[4 * h] + blk, cur.c[4 * h + 1] + blk, cur.c[4 * h + 2] + blk, cur.c[4 * h + 3] + blk);
			} else {
#pragma unroll
				for (int i = 0; i < 3; ++i)
					if (c + i < ncol) lds_dma1<X4>(voff, d + (unsigned)i * (unsigned)STRIDE, cur.c[4 * h + i] + blk);
			}
		}
		cur = nxt;
	}
}

__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void lds_dma_wait_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

} // namespace anofox
