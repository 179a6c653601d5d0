// The control word of the resident block kernel (doppler_amd/csrc/dpx_types.h, BlockCtl) must prove itself: a 16-byte read
// that mixes dwords of two successive writes of a slot must never pass for a rung doorbell with the wrong payload.
// Enumerates every mix of (ticket, payload, ticket) dwords taken from an old and a new write, for tickets around the
// wrap of the 9-bit tag and of the 32-bit ticket, and payload pairs that differ in every field.  Host only.
#include <stdio.h>
#include <initializer_list>
#include <stdint.h>
#include "../../doppler_amd/csrc/dpx_types.h"

int main()
{
    using namespace dpx;
    long checked = 0, accepted_mixed = 0;
    const uint32_t tickets[] = {1, 2, 3, 4, 5, 508, 509, 510, 511, 512, 513, 1021, 1024, 0x7ffffffeu, 0xfffffff0u, 0xfffffffau, kDoorExit};
    const uint32_t ns[] = {0, 1, 2047, 2048, 2049, 8191, 8192};
    const uint32_t segs[] = {1, 2, 16};
    for (uint32_t t_old : tickets)
        for (uint32_t step : {4u, 8u, 1u}) {               // consecutive tickets of a slot differ by kResidentSlots; 1: paranoia
            uint32_t t_new = t_old + step;
            if (t_new == 0 || t_new == kDoorExit) t_new = 1;
            for (uint32_t n0 : ns) for (uint32_t n1 : ns) for (uint32_t s0 : segs) for (uint32_t s1 : segs)
                for (uint32_t inst0 = 0; inst0 < 8; inst0 += 3) for (uint32_t inst1 = 0; inst1 < 8; inst1 += 5) {
                    const uint32_t old_w[3] = {t_old, ctl_payload(t_old, n0, s0, 0, inst0), t_old};
                    const uint32_t new_w[3] = {t_new, ctl_payload(t_new, n1, s1, 1, inst1), t_new};
                    for (int mix = 0; mix < 8; ++mix) {
                        const uint32_t w0 = (mix & 1) ? new_w[0] : old_w[0], w1 = (mix & 2) ? new_w[1] : old_w[1], w3 = (mix & 4) ? new_w[2] : old_w[2];
                        ++checked;
                        if (!ctl_word_valid(w0, w1, w3)) continue;
                        // accepted: the word must be one of the two writes as a whole
                        const bool is_old = w0 == old_w[0] && w1 == old_w[1] && w3 == old_w[2];
                        const bool is_new = w0 == new_w[0] && w1 == new_w[1] && w3 == new_w[2];
                        if (!is_old && !is_new) {
                            if (++accepted_mixed < 10) printf("MIXED WORD ACCEPTED: tickets %u -> %u, mix %d\n", t_old, t_new, mix);
                        }
                    }
                    // and the fields come back
                    if ((new_w[1] & 0x3fffu) != n1 || ((new_w[1] >> 14) & 0x1fu) != s1 || ((new_w[1] >> 19) & 1u) != 1u || ((new_w[1] >> 20) & 7u) != inst1) {
                        printf("payload fields do not round-trip\n");
                        return 1;
                    }
                }
        }
    printf("%ld words checked, %ld mixed words accepted\n", checked, accepted_mixed);
    return accepted_mixed == 0 ? 0 : 1;
}
