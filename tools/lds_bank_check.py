#!/usr/bin/env python
"""Host-side check of the LDS layouts of attn_mfma2.hip against the bank model of /opt/skills/guides/MI355X_MICROARCH.md (LDS table):
a wave64 ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, and the same + 32), a
ds_read_b64_tr_b16 in two groups of 32 lanes; a group costs as many LDS cycles as the most loaded of the 64 four-byte banks has
distinct addresses.  Prints, per head size, the cycles of the K fragment reads (ideal 4) and of the transpose reads of V
(ideal 2) under the swizzle key a2_key<NV> of the kernel, and of the padded register-staged layouts.   python tools/lds_bank_check.py"""

GROUPS_B128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS_B128 += [[x + 32 for x in g] for g in GROUPS_B128]
GROUPS_TR = [list(range(32)), list(range(32, 64))]


def cycles(groups, addr_fn, dwords):
    total = 0
    for g in groups:
        bank = {}
        for lane in g:
            a = addr_fn(lane)
            for b in range(dwords):
                bank.setdefault((a // 4 + b) % 64, set()).add(a)
        total += max(len(v) for v in bank.values())
    return total


def a2_key(nv, row):  # attn_mfma2.hip a2_key<NV>
    return row & (nv - 1) if nv in (8, 16) else (row >> 1) & 3


def main():
    for dh in (32, 64, 96, 128):
        nv, row_bytes = dh // 8, dh * 2
        k = max(cycles(GROUPS_B128, lambda l: (l & 15) * row_bytes + (((ks * 4 + (l >> 4)) ^ a2_key(nv, l & 15)) * 16), 4) for ks in range(dh // 32))
        tr = 0
        for eb in range(dh // 16):
            for j in range(2):
                def addr(l, eb=eb, j=j):
                    c, g = l & 15, l >> 4
                    key = j * 32 + g * 4 + (c >> 2)
                    return key * row_bytes + (((eb * 2 + ((c & 3) >> 1)) ^ a2_key(nv, key)) << 4) + ((c & 1) << 3)
                tr = max(tr, cycles(GROUPS_TR, addr, 2))
        pads = {pad: max(cycles(GROUPS_B128, lambda l: (l & 15) * (row_bytes + pad) + (ks * 4 + (l >> 4)) * 16, 4) for ks in range(dh // 32))
                for pad in (16, 32)}
        print(f"dh {dh:3d}: swizzled K fragment reads {k} cycles (ideal 4), transpose reads of V {tr} (ideal 2); "
              f"register-staged rows padded by 16 / 32 bytes: {pads[16]} / {pads[32]}")


if __name__ == "__main__":
    main()
