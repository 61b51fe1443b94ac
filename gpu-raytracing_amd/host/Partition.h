// Partition.h -- how a frame is cut across the GPUs of one node (the reference traces on one GPU,
// main.cu:169, so there is no reference counterpart).  Pure arithmetic, no device code: unit-tested on the CPU through
// librt_host.so (tests/test_host_mirror.py) against gpu-raytracing_amd/sharding.py, which the Python harness uses.
//
//   bands  : device d of P renders the contiguous rows [d*H/P, (d+1)*H/P) in place in its own full-frame buffer.
//   strips : interleaved strips of kStripRows rows; strip s belongs to device s mod P; a device's strips are rendered
//            compactly (its j-th strip = global strip d + j*P) by rt_trace_strips and de-interleaved on device 0.
#pragma once
#include <cstdint>

constexpr unsigned kStripRows = 8;            // one tile row of the tracer (a wave is an 8x8 pixel tile)
constexpr double kImbalanceLimit = 1.15;      // bands -> strips when max / mean band cost exceeds this

enum class Partition { kBands = 0, kStrips = 1, kAuto = 2 };

struct RowBand { unsigned y0, y1; };
inline RowBand BandOf(unsigned height, unsigned devices, unsigned d)
{
    return RowBand{(unsigned)(((uint64_t)d * height) / devices), (unsigned)(((uint64_t)(d + 1) * height) / devices)};
}

inline unsigned NumStrips(unsigned height) { return (height + kStripRows - 1) / kStripRows; }
// strips in every device's compact buffer (the same for all, so the pieces that travel have one size)
inline unsigned StripsPerDevice(unsigned height, unsigned devices) { return (NumStrips(height) + devices - 1) / devices; }
// strips device d really owns (the last devices may own one fewer)
inline unsigned StripsOwned(unsigned height, unsigned devices, unsigned d)
{
    const unsigned s = NumStrips(height);
    return d < s ? (s - d + devices - 1) / devices : 0u;
}
inline unsigned CompactRows(unsigned height, unsigned devices) { return StripsPerDevice(height, devices) * kStripRows; }
// rows of global strip s that lie inside the frame (the last strip may be cut by the frame's edge)
inline unsigned StripRowsInFrame(unsigned height, unsigned s)
{
    const unsigned y = s * kStripRows;
    return y >= height ? 0u : (height - y < kStripRows ? height - y : kStripRows);
}

// 'strips' when the per-band costs (any unit) are uneven: max / mean > limit
inline Partition ChoosePartition(const double* band_costs, unsigned devices, double limit = kImbalanceLimit)
{
    double sum = 0, mx = 0;
    for (unsigned d = 0; d < devices; d++) { sum += band_costs[d]; mx = band_costs[d] > mx ? band_costs[d] : mx; }
    const double mean = devices ? sum / devices : 0.0;
    if (devices < 2 || mean <= 0.0) return Partition::kBands;
    return mx / mean > limit ? Partition::kStrips : Partition::kBands;
}
