"""sha256 (16 hex digits) over the HIP sources of the library, in a fixed order:
what a committed PMC summary was collected on.  bench.py compares it with the
tree it runs from (`traffic_stale`).  python tools/csrc_hash.py"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'tensorflow-wavenet_amd', 'csrc')


def csrc_hash():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith(('.hip', '.h')):
            h.update(name.encode())
            with open(os.path.join(CSRC, name), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(csrc_hash())
