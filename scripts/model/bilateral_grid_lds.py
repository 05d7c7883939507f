"""LDS cycles per workgroup of bg_blur_slice's three LDS stages under the gfx950 banking rules (lds_banks.py), for the round-4
layout ([z][row][cell] tiles, a wave = 64 pixels of one row) and the round-5 one (z-innermost tiles, a 32-lane half = 16 pixels
x 2 rows).  `python scripts/model/bilateral_grid_lds.py`"""
import numpy as np
from lds_banks import cycles

FCX, FCY = 10, 6


def old_layout(ZD=12, noise=True, seed=0):
    bz_row, bx_row, by_row = FCX + 4, FCX, FCX
    bz_plane, bx_plane, by_plane = (FCY + 4) * bz_row, (FCY + 4) * bx_row, FCY * by_row
    cyc = 0
    n = ZD * (FCY + 4) * FCX
    for w0 in range(0, -(-n // 256) * 256, 64):
        e = np.arange(w0, w0 + 64); act = e < n
        if not act.any(): continue
        z = e // ((FCY + 4) * FCX); rem = e % ((FCY + 4) * FCX); j = rem // FCX; i = rem % FCX
        for d in range(5):
            cyc += cycles(8 * (z * bz_plane + j * bz_row + i + d), "r64", act)
        cyc += cycles(8 * (z * bx_plane + j * bx_row + i), "w64", act)
    n = ZD * FCY * FCX
    for w0 in range(0, -(-n // 256) * 256, 64):
        e = np.arange(w0, w0 + 64); act = e < n
        if not act.any(): continue
        z = e // (FCY * FCX); rem = e % (FCY * FCX); j = rem // FCX; i = rem % FCX
        for d in range(5):
            cyc += cycles(8 * (z * bx_plane + (j + d) * bx_row + i), "r64", act)
        cyc += cycles(8 * (z * by_plane + j * by_row + i), "w64", act)
    rng = np.random.default_rng(seed)
    for y in range(32):
        x = np.arange(64); xi = x // 8; yi = y // 8
        zi = rng.integers(0, 11, 64) if noise else np.clip((5 + 2 * np.sin(x / 40.0 + y / 9.0)).astype(int), 0, 10)
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    cyc += cycles(8 * ((zi + dz) * by_plane + (yi + dy) * by_row + xi + dx), "r64")
    return cyc


def new_layout(ZP=12, noise=True, seed=0, ox0=0):
    NBX, NBY = (FCY + 4) * FCX * ZP, FCY * FCX * ZP
    cyc = 0
    for w0 in range(0, -(-NBX // 256) * 256, 64):
        e = np.arange(w0, w0 + 64); act = e < NBX
        if not act.any(): continue
        z = e % ZP; c = e // ZP; j = c // FCX; i = c % FCX
        for d in range(5):
            cyc += cycles(8 * ((j * (FCX + 4) + i + d) * ZP + z), "r64", act)
        cyc += cycles(8 * e, "w64", act)
    for w0 in range(0, -(-NBY // 256) * 256, 64):
        e = np.arange(w0, w0 + 64); act = e < NBY
        if not act.any(): continue
        for d in range(5):
            cyc += cycles(8 * (e + d * FCX * ZP), "r64", act)
        cyc += cycles(8 * e, "w64", act)
    rng = np.random.default_rng(seed)
    lane = np.arange(64)
    for wave in range(4):
        pcol = (wave & 1) * 32 + ((lane >> 5) << 4) + (lane & 15)
        prow = ((wave >> 1) << 1) + ((lane >> 4) & 1)
        for k in range(8):
            y = prow + 4 * k
            ax = ox0 + pcol
            xi = ax // 8 - ox0 // 8; yi = y // 8
            zi = rng.integers(0, 11, 64) if noise else np.clip((5 + 2 * np.sin(pcol / 40.0 + y / 9.0)).astype(int), 0, 10)
            for dz in (0, 1):
                for dy in (0, 1):
                    for dx in (0, 1):
                        cyc += cycles(8 * (((yi + dy) * FCX + xi + dx) * ZP + zi + dz), "r64")
    return cyc


if __name__ == "__main__":
    for noise in (True, False):
        print("noise " if noise else "smooth", "round 4 layout:", old_layout(noise=noise), " round 5 layout: ZP=12", new_layout(12, noise), " ZP=16", new_layout(16, noise),
              " ZP=12, output origin x = 3:", new_layout(12, noise, ox0=3))
