#define BAND_U float
#include "launch_band.inc"

namespace viprs {

int band_ring_panels(const viprs_plan* P) {
    int rp = 4;
    while (rp < P->max_band_panels) rp *= 2;
    return rp;
}

}  // namespace viprs
