"""Generates tests/golden/fill_uniform_seed1.npy with an independent pure-Python splitmix64
(the input stream of SURVEY.md §8(d)); the C oracle and the device fill kernel must match it."""
import os
import numpy as np

M = (1 << 64) - 1


def splitmix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def fill(n, seed, scale=1.0):
    base = (seed * 0x9E3779B97F4A7C15) & M
    return np.array([scale * ((splitmix64((base + i) & M) >> 11) * 2.0 ** -53) for i in range(n)])


if __name__ == "__main__":
    np.save(os.path.join(os.path.dirname(__file__), "fill_uniform_seed1.npy"), fill(64, 1))
