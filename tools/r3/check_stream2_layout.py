"""Exhaustive check of the LDS layout of topk_stream2_kernel (uniir_amd/csrc/topk.hip), on the host, no GPU needed:
  1. the LDS-DMA source offsets (vb0 / vb1 + instruction offset) and the fragment read offsets (la0 / la1 + immediate) agree:
     the 16 bytes lane (li, lg) reads for k-step s are row li, dims 32 s + 8 lg .. + 8 of the tile;
  2. every DMA instruction's 64 lanes fetch 8 whole 128-byte lines (8 consecutive lanes = one line);
  3. ds_read_b128 is bank-conflict free: within each of its four 16-lane service groups (MI355X guide, LDS table) the lanes hit
     16 distinct 16-byte slots modulo 256 bytes."""
ROW_BYTES = 1536


def dma_src(j, lane, half):
    r8, c8 = lane >> 3, lane & 7
    row = 8 * (j & 1) + r8
    vb = row * ROW_BYTES + ((c8 ^ ((row >> 1) & 7)) << 4)
    return vb + (j >> 1) * 128 + half * 768          # byte offset inside the 16-row tile


def frag_addr(lane, sh):
    li, lg = lane & 15, lane >> 4
    g = (li >> 1) & 7
    base = (li >> 3) * 1024 + (li & 7) * 128 + ((((4 if sh & 1 else 0) + lg) ^ g) << 4)
    return base + (sh >> 1) * 2048


GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]

for half in (0, 1):
    lds = {}
    for j in range(12):
        lines = set()
        for lane in range(64):
            src = dma_src(j, lane, half)
            lds[j * 1024 + lane * 16] = src
            lines.add((src // ROW_BYTES, (src % ROW_BYTES) // 128))
            assert src % 16 == 0
        assert len(lines) == 8, (j, lines)                      # 8 full lines per instruction
        for grp in range(8):                                     # 8 consecutive lanes share one line
            assert len({(dma_src(j, 8 * grp + c, half) // ROW_BYTES, (dma_src(j, 8 * grp + c, half) % ROW_BYTES) // 128)
                        for c in range(8)}) == 1
    assert len(lds) == 768 and len(set(lds.values())) == 768     # a bijection onto the half-tile's 768 chunks
    for sh in range(12):
        s = 12 * half + sh
        for lane in range(64):
            li, lg = lane & 15, lane >> 4
            want = li * ROW_BYTES + (32 * s + 8 * lg) * 2
            assert lds[frag_addr(lane, sh)] == want, (half, sh, lane)
        for grp in GROUPS:
            slots = {(frag_addr(l, sh) % 256) // 16 for l in grp}
            assert len(slots) == 16, (sh, sorted(slots))
print("stream2 layout ok: source/fragment maps agree, full-line DMA, conflict-free ds_read_b128")
