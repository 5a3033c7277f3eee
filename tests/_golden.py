"""Helpers to read the committed golden fixtures (tests/golden/*.npz)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'))


def jload(arr):
    return json.loads(bytes(arr.tobytes()).decode())


def meta_from_arrays(n, cyx, area, score):
    """dense golden arrays -> list[dict[id -> {center_yx, area, score}]]"""
    out = []
    for b in range(len(n)):
        out.append({i + 1: {'center_yx': (int(cyx[b, i, 0]), int(cyx[b, i, 1])),
                            'area': int(area[b, i]),
                            'score': float(score[b, i])}
                    for i in range(int(n[b]))})
    return out


def ids_from_arrays(n, k, v):
    return [{int(k[b, i]): int(v[b, i]) for i in range(int(n[b]))}
            for b in range(len(n))]
