"""CPU replay of the device schedule with the kernel's exact memory semantics
(kernels.hip cell_kernel): rows gathered into an LDS image, the rows of step
t+1 read BEFORE the rows of step t are written back, flagged q rows forwarded in
registers, run-mode q rows resident for the whole run.  Arithmetic is the
oracle's mfo_sgd_update.  If the scheduler ever breaks a hazard rule the replay
reads a stale row and diverges from the sequential oracle, without needing a GPU.
"""
import numpy as np


def _decode(x, L):
    return (int(x) & 0xFFFF) // L, ((int(x) >> 16) & 0x7FFF) // L, (int(x) >> 31) & 1


def replay_epoch(oracle, P, Q, k, lr, lam, sched, B, W, G, L):
    cells, rows, subs, entries = sched
    for rd in range(B):
        for b in range(B):
            # a cell is a chain of chunks (descriptor word 4 = next chunk, 0 = last), run back to back
            chain, cell = [], b * B + (b + rd) % B
            while True:
                chain.append(cell)
                cell = int(cells[cell][4])
                if cell == 0:
                    break
                assert cell >= B * B, "chunk links must point behind the first-chunk table"
            for cell in chain:
                _replay_chunk(oracle, P, Q, k, lr, lam, sched, cell, W, G, L)


def _replay_chunk(oracle, P, Q, k, lr, lam, sched, cell, W, G, L):
    cells, rows, subs, entries = sched
    if True:
        if True:
            row_off, ent_off, n_steps, nuni = (int(v) for v in cells[cell][:4])
            n_steps &= 0x7FFFFFFF  # bit 31: "carries a run" hint
            nu, ni = nuni & 0xFFFF, nuni >> 16
            nrows = nu + ni
            if nrows == 0:
                return
            ids = rows[row_off:row_off + nrows]
            lds = np.zeros((nrows + 2 * G, k), np.float32)
            lds[:nu] = P[ids[:nu]]
            lds[nu:nrows] = Q[ids[nu:]]

            def ent(step, g):
                return entries[(ent_off + step) * G + g]

            for s in range(W):
                touched = {}
                for w in range(W):
                    off, n = (int(v) for v in subs[cell * W * W + s * W + w])
                    ng, nr = n & 0xFFFF, n >> 16
                    off, nsolo = off & 0xFFFF, off >> 16
                    # ---- general steps --------------------------------------------------
                    if ng > 0:
                        cur = []
                        for g in range(G):
                            pa, qa, _ = _decode(ent(off, g)[0], L)
                            cur.append([pa, qa, lds[pa].copy(), lds[qa].copy()])
                        for t in range(ng):
                            nxt = []
                            for g in range(G):  # prefetch of step t+1, before this step's stores
                                pa, qa, fwd = _decode(ent(off + t + 1, g)[0], L)
                                nxt.append([pa, qa, lds[pa].copy(), lds[qa].copy(), fwd])
                            new = []
                            real_rows = []
                            for g in range(G):
                                pa, qa, p, q = cur[g]
                                r = float(ent(off + t, g)[1:2].view(np.float32)[0])
                                p2, q2 = p.copy(), q.copy()
                                oracle.sgd_update(p2, q2, r, lr, lam)
                                new.append((p2, q2))
                                if pa < nrows:
                                    real_rows += [pa, qa]
                                    for row in (pa, qa):
                                        assert touched.setdefault(row, w) == w, "row shared by two waves in a sub-round"
                            assert len(real_rows) == len(set(real_rows)), "two slots of a step share a row"
                            for g in range(G):
                                lds[cur[g][0]] = new[g][0]
                                lds[cur[g][1]] = new[g][1]
                            for g in range(G):
                                pa, qa, pn, qn, fwd = nxt[g]
                                cur[g] = [pa, qa, pn, new[g][1] if fwd else qn]
                    # ---- run steps ------------------------------------------------------
                    if nr > 0:
                        base = off + ng
                        rqa, rq, curp = [], [], []
                        for g in range(G):
                            pa, qa, _ = _decode(ent(base, g)[0], L)
                            rqa.append(qa)
                            rq.append(lds[qa].copy())
                            curp.append([pa, lds[pa].copy()])
                        assert len(set(rqa)) == len(rqa)
                        for t in range(nr):
                            nxtp = []
                            for g in range(G):
                                pa, qa, _ = _decode(ent(base + t + 1, g)[0], L)
                                nxtp.append([pa, lds[pa].copy()])
                            real_rows = []
                            for g in range(G):
                                pa, qa, idle = _decode(ent(base + t, g)[0], L)
                                assert pa == curp[g][0]
                                if idle:
                                    # the kernel does not test the flag: it relies on r = 0, ce = 1
                                    assert ent(base + t, g)[1] == 0 and ent(base + t, g)[2] == 0
                                    assert ent(base + t, g)[3:4].view(np.float32)[0] == 1.0
                                    continue
                                assert qa == rqa[g], "run entry changes the resident item"
                                r = float(ent(base + t, g)[1:2].view(np.float32)[0])
                                assert ent(base + t, g)[2:3].view(np.float32)[0] == np.float32(lr) * np.float32(r)
                                assert ent(base + t, g)[3:4].view(np.float32)[0] == np.float32(1.0) - np.float32(lr) * np.float32(lam)
                                p2 = curp[g][1].copy()
                                oracle.sgd_update(p2, rq[g], r, lr, lam)
                                lds[pa] = p2
                                real_rows.append(pa)
                                for row in (pa, qa):
                                    assert touched.setdefault(row, w) == w, "row shared by two waves in a sub-round"
                            assert len(real_rows) == len(set(real_rows))
                            curp = nxtp
                        for g in range(G):
                            lds[rqa[g]] = rq[g]
                    # ---- solo records: [header][record 0..n-1][terminator], 16 bytes each ------------
                    if nsolo > 0:
                        first = (ent_off + off + ng + nr + 2) * G  # two idle steps (look-ahead padding) first
                        hdr = entries[first]
                        pa, qa, _ = _decode(hdr[0], L)  # record = {slots of the next step, mailbox, lr * r, r}
                        assert qa < nrows and hdr[1] == 0
                        q = lds[qa].copy()  # the helper wave reads it once and stores it at the end
                        seen = set()
                        c = np.float32(1.0) - np.float32(lr) * np.float32(lam)
                        for t in range(nsolo):
                            rec = entries[first + 1 + t]
                            r = float(rec[3:4].view(np.float32)[0])
                            assert rec[2:3].view(np.float32)[0] == np.float32(lr) * np.float32(r)
                            assert rec[1] == 0xFFFFFFFF, "mailbox must start out empty"
                            assert pa < nu and pa not in seen, "solo run: a user row twice (its helper stores p rows late)"
                            seen.add(pa)
                            for row in (pa, qa):
                                assert touched.setdefault(row, w) == w, "row shared by two waves in a sub-round"
                            p2 = lds[pa].copy()
                            oracle.sgd_update(p2, q, r, lr, lam)
                            lds[pa] = p2
                            pa, qn, _ = _decode(rec[0], L)
                            assert qn == qa
                        assert pa == nrows, "the address behind the last solo step must be the zero row"
                        term = entries[first + 1 + nsolo]
                        assert _decode(term[0], L)[0] == nrows and term[2] == 0 and term[3] == 0 and term[1] == 0xFFFFFFFF
                        lds[qa] = q
            assert not lds[nrows:].any(), "an idle slot dirtied the all-zero rows"
            P[ids[:nu]] = lds[:nu]
            Q[ids[nu:]] = lds[nu:nrows]
