#!/usr/bin/env python3
"""Soak test of the frames-in-flight pipe: N single-frame submits cycling over a few frame sets with dynamic batching, two threads (submit / wait); every result must be
byte-identical to the first result seen for the same (frame set, previous frame set) -- a race between lanes, passes or threads shows up as a hash that changes.
Usage: python tools/soak_pipe.py [n_submits=20000] [lanes=4] [coalesce=4] [depth=2]"""
import hashlib, os, queue, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H, W, CAP = 480, 640, 200


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    coal = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    depth = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    from d2slam_amd import api, netvlad as nvm
    from d2slam_amd.synth import synth_stereo
    from d2slam_amd.weights import synthetic_superpoint_weights
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2, precision=api.PREC_F32_WINO))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5)); fe.load_netvlad(nvm.synthetic_netvlad_weights())
    NS = 5
    sets = []
    for s in range(NS):
        l, r = synth_stereo(H, W, seed=40 + s % 3)
        sh = (s % 3, (2 * s) % 5)
        sets.append((np.ascontiguousarray(np.roll(l, sh, (0, 1))[None]), np.ascontiguousarray(np.roll(r, sh, (0, 1))[None])))
    pipe = api.StereoPipe(fe, lanes=lanes, frames=1, width=W, height=H, cap=CAP, netvlad=True, ratio=0.8, coalesce=coal, coalesce_depth=depth)
    q = queue.Queue(maxsize=lanes * coal)
    seen, bad, errors = {}, [], []

    def producer():
        try:
            for i in range(n):
                s = sets[i % NS]
                q.put((i, pipe.submit(s[0], s[1])))
        except Exception as e:      # noqa: BLE001
            errors.append(e)
        q.put(None)

    def consumer():
        try:
            while True:
                it = q.get()
                if it is None:
                    return
                i, t = it
                o = pipe.wait(t)
                hsh = hashlib.blake2b(digest_size=16)
                nk = o["n_kp"]
                hsh.update(nk.tobytes())
                for im in range(2):
                    k = int(nk[im]); hsh.update(o["kps_xy"][im, :k].tobytes()); hsh.update(o["scores"][im, :k].tobytes()); hsh.update(o["desc"][im, :k].tobytes())
                hsh.update(o["netvlad"].tobytes())
                for pre in ("lr", "prev"):
                    m = int(o[pre + "_n"][0]); hsh.update(o[pre + "_n"].tobytes())
                    hsh.update(o[pre + "_q"][0, :m].tobytes()); hsh.update(o[pre + "_t"][0, :m].tobytes()); hsh.update(o[pre + "_dist"][0, :m].tobytes())
                key = (i % NS, (i - 1) % NS if i else -1)
                d = hsh.hexdigest()
                if seen.setdefault(key, d) != d:
                    bad.append((i, key))
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    t0 = time.perf_counter()
    th = [threading.Thread(target=producer), threading.Thread(target=consumer)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("soak: %d submits in %.1f s (%.0f stereo fps incl. hashing), %d distinct (frame, previous) keys, %d mismatches, %d errors"
          % (n, dt, n / dt, len(seen), len(bad), len(errors)))
    if errors:
        print(errors[:3])
    pipe.close(); fe.close()
    sys.exit(1 if bad or errors else 0)


if __name__ == "__main__":
    main()
