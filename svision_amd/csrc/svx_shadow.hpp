// svx_shadow.hpp -- register-file room for a co-resident wave of another kernel.
//
// The convolutions and fc6 / fc7 are bound by the matrix pipe; the BGZF inflate kernel (svx_inflate.hip) is bound by vector /
// scalar issue and latency.  A SIMD runs both kinds at once (separate pipes: MI355X_MICROARCH.md "Wave scheduling") if
// both fit its 512-entry register file -- but a matrix kernel of ~120 VGPRs takes four wave slots per SIMD, all of the file,
// and a 64-ms inflate wave never gets in beside it.  SVX_SHADOW_ROOM() raises the kernel's ALLOCATION to
// SVX_SHADOW_VGPRS registers (an asm clobber: no instruction, no spill) so that at most two of its waves share a SIMD --
// their operating point anyway (svx_conv.hip: "two small waves per SIMD") -- and 160 registers stay free for one
// inflate wave; SVX_SHADOW_PRIO gives the matrix waves the issue priority over the (older, long-lived) inflate wave.
#pragma once
#ifdef SVX_SHADOW_VGPRS
#define SVX_SHADOW_STR2(x) #x
#define SVX_SHADOW_STR(x) SVX_SHADOW_STR2(x)
#ifdef SVX_SHADOW_PRIO
#define SVX_SHADOW_ROOM() do { asm volatile("; room for a co-resident wave" ::: "v" SVX_SHADOW_STR(SVX_SHADOW_VGPRS)); __builtin_amdgcn_s_setprio(SVX_SHADOW_PRIO); } while (0)
#else
#define SVX_SHADOW_ROOM() asm volatile("; room for a co-resident wave" ::: "v" SVX_SHADOW_STR(SVX_SHADOW_VGPRS))
#endif
#else
#define SVX_SHADOW_ROOM() do {} while (0)
#endif
