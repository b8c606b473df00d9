// Launchers of conv_direct.hip (the thin-GEMM layers: 8 input channels -> 64, or <= 8 outputs), called by the tile
// selection of o2m_conv2d_fwd in conv_igemm.hip.  Internal to libo2m_hip.so.
#pragma once
#include "common.h"

namespace o2m_direct {
bool stem8_ok(const o2m_conv_desc& d);         // Ci == 8, Co == 64, 4 x 4 or 7 x 7, plain epilogue (bias, activation, IN partials)
int stem8_stats_rows(const o2m_conv_desc& d);  // consecutive output pixels per InstanceNorm partial (0: cannot emit them)
int launch_stem8(const o2m_conv_desc& d, hipStream_t s);
bool fewout_ok(const o2m_conv_desc& d);        // Co == 8, Ci % 32 == 0, 4 x 4, zero padding
int launch_fewout(const o2m_conv_desc& d, hipStream_t s);
}  // namespace o2m_direct
