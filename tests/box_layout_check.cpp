// The REF sampler's table layout (csrc/ddgi_types.h: box_slots / box_slot_xyz / box_slot_to_slab_slot) as pure host code: for a set of grids —
// odd counts included — the probes map one to one into [0, box_slots), the inverse gives the slab slot back, the padding slots say -1, and the
// 8 probes of an aligned 2x2x2 brick share one group of 8 consecutive slots.  Built and run by tests/test_abi.py (no GPU).
#include "ddgi_types.h"

#include <cstdio>
#include <vector>

int main()
{
    const int grids[][3] = {{2, 2, 2}, {8, 8, 8}, {32, 16, 32}, {5, 3, 7}, {1, 1, 1}, {9, 4, 1}, {64, 32, 16}, {3, 1, 2}};
    for (const auto& g : grids)
    {
        const int cx = g[0], cy = g[1], cz = g[2];
        const uint32_t n = ddgi::box_slots(cx, cy, cz);
        if (n % 8u != 0u || n < static_cast<uint32_t>(cx * cy * cz)) return std::printf("box_slots(%d,%d,%d) = %u\n", cx, cy, cz, n), 1;
        std::vector<int> seen(n, -1);
        for (int z = 0; z < cz; ++z)
            for (int y = 0; y < cy; ++y)
                for (int x = 0; x < cx; ++x)
                {
                    const uint32_t b = ddgi::box_slot_xyz(cx, cy, x, y, z);
                    const int slab = (z * cy + y) * cx + x;
                    if (b >= n || seen[b] != -1) return std::printf("(%d,%d,%d) of %dx%dx%d -> slot %u: out of range or taken\n", x, y, z, cx, cy, cz, b), 1;
                    seen[b] = slab;
                    if (ddgi::box_slot_to_slab_slot(cx, cy, cz, b) != slab) return std::printf("slot %u does not lead back to slab slot %d\n", b, slab), 1;
                    // separable: a slot is the sum of one term per axis (the sampler adds them)
                    if (b != ddgi::box_slot_xyz(cx, cy, x, 0, 0) + ddgi::box_slot_xyz(cx, cy, 0, y, 0) + ddgi::box_slot_xyz(cx, cy, 0, 0, z)) return std::printf("slot %u is not the sum of its axes' terms\n", b), 1;
                    // a brick's probes share a group of 8 slots
                    if (b / 8u != ddgi::box_slot_xyz(cx, cy, x & ~1, y & ~1, z & ~1) / 8u) return std::printf("(%d,%d,%d) is not in its brick's line\n", x, y, z), 1;
                }
        for (uint32_t b = 0; b < n; ++b)
            if (seen[b] == -1 && ddgi::box_slot_to_slab_slot(cx, cy, cz, b) != -1) return std::printf("padding slot %u of %dx%dx%d maps to a probe\n", b, cx, cy, cz), 1;
    }
    std::printf("ok\n");
    return 0;
}
