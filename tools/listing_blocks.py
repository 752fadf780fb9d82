"""Basic-block summary of a device listing: instructions, MFMAs, scratch / lane-spill traffic,
barriers and memory instructions per block (where does the compiler spill?).

    python tools/listing_blocks.py <listing.s> [kernel-name-prefix]
"""
import re
import sys


def blocks_of(lines):
    blocks, cur = [], None
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m or cur is None:
            cur = dict(name=m.group(1) if m else 'entry', line=i + 1, n=0, mfma=0, sst=0, sld=0, wl=0, rl=0,
                       bar=0, ds=0, buf=0, glob=0)
            blocks.append(cur)
            if m:
                continue
        s = l.strip()
        if not s or s[0] in ';.':
            continue
        cur['n'] += 1
        for key, pat in (('mfma', 'v_mfma'), ('sst', 'scratch_store'), ('sld', 'scratch_load'),
                         ('wl', 'v_writelane'), ('rl', 'v_readlane'), ('bar', 's_barrier')):
            if pat in s:
                cur[key] += 1
        for key, pat in (('ds', 'ds_'), ('buf', 'buffer_'), ('glob', 'global_')):
            if s.startswith(pat):
                cur[key] += 1
    return blocks


def main():
    text = open(sys.argv[1]).read().split('\n')
    prefix = sys.argv[2] if len(sys.argv) > 2 else None
    if prefix:
        start = next(i for i, l in enumerate(text) if l.startswith(prefix) and l.rstrip().endswith(':') or
                     (l.startswith(prefix) and ':' in l))
        end = next(i for i in range(start, len(text)) if text[i].strip().startswith('.amdhsa_kernel') or
                   text[i].startswith('.Lfunc_end'))
        text = text[start:end]
    for b in blocks_of(text):
        if b['n'] > 15 or b['sst'] or b['sld'] or b['bar']:
            print(' '.join('%s=%s' % kv for kv in b.items()))


if __name__ == '__main__':
    main()
