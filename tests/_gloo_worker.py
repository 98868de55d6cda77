"""Worker for test_world_size_2_gloo_sharded_path (launched by torch.distributed.run, backend gloo, CPU).

The compute engine on CPU is the oracle (allowed: this is test code); what is under test is the package's
host-side sharding logic: shard_rows, allreduce_sum_ with the [grad ; f] payload convention, that the
sharded operator drives the iteration to the unsharded answer, and the arithmetic protocols of the column-sharded sweep and of the
row-team sweep (per-column partial dots summed in rank order) restated on the CPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import proxgrad_oracle as o  # noqa: E402
import proximalalgorithms.jl_amd as pa  # noqa: E402


class ShardedOracleLS:
    """LeastSquares over a row shard with the payload convention of csrc/pg_gemv.hip: buf = [grad ; f]."""

    def __init__(self, A_loc, b_loc):
        self.loc = o.LeastSquares(A_loc, b_loc)
        self.calls = 0

    def value_and_gradient(self, x):
        f, g = self.loc.value_and_gradient(x)
        buf = torch.from_numpy(np.concatenate([g, np.array([f], dtype=g.dtype)]))
        pa.allreduce_sum_(buf)
        self.calls += 1
        out = buf.numpy()
        return out[-1], out[:-1].copy()


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    m, n = 96, 160
    for dtype in (np.float32, np.float64):
        A, b, _ = o.synthetic_lasso(m, n, seed=0, dtype=dtype)
        lam = dtype(0.1) * dtype(np.max(np.abs(A.T @ b)))
        off, cnt = pa.shard_rows(m, world, rank)
        A_loc = o.synthetic_matrix(cnt, n, seed=0, dtype=dtype, row_offset=off, m_global=m)
        assert np.array_equal(A_loc, A[off:off + cnt])  # shards regenerate the same entries
        f_sh = ShardedOracleLS(A_loc, b[off:off + cnt])
        x = np.random.default_rng(1).standard_normal(n).astype(dtype)
        fs, gs = f_sh.value_and_gradient(x)
        ff, gf = o.LeastSquares(A, b).value_and_gradient(x)
        tol = 1e-5 if dtype == np.float32 else 1e-12
        assert abs(fs - ff) <= tol * abs(ff) and np.max(np.abs(gs - gf)) <= tol * np.linalg.norm(gf)
        z_sh, k_sh = o.fast_forward_backward(tol=1e-5, x0=np.zeros(n, dtype), f=f_sh, g=o.NormL1(lam))
        z, k = o.fast_forward_backward(tol=1e-5, x0=np.zeros(n, dtype), f=o.LeastSquares(A, b), g=o.NormL1(lam))
        assert abs(k_sh - k) <= 3, (k_sh, k)
        assert np.max(np.abs(z_sh - z)) <= (1e-4 if dtype == np.float32 else 1e-9)
        # replicated vectors stay replicated: both ranks hold the same iterate
        t = torch.from_numpy(z_sh.astype(np.float64).copy())
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        assert torch.equal(gathered[0], gathered[1])
    # ---- column shards: the single-sweep protocol of csrc/pg_gemv.hip (ls_fused_pass_t, column mode) on the CPU ----
    # every rank holds A[:, J_p] and the J_p slices of the n-vectors; per iteration ONE all-reduce of
    # [partial of A v (m) ; 8 * world scalar slots (hi / lo pairs)]; the slots turn the SUM into an all-gather (sum, max, sum, sum)
    for dtype in (np.float32, np.float64):
        A, b, _ = o.synthetic_lasso(m, n, seed=0, dtype=dtype)
        lam = dtype(0.1) * dtype(np.max(np.abs(A.T @ b)))
        Lf = dtype(1.05 * np.linalg.norm(A.astype(np.float64), 2) ** 2)
        coff, ccnt = pa.shard_cols(n, world, rank)
        A_loc = A[:, coff:coff + ccnt]
        gamma = dtype(1) / Lf
        ref = iter(o.FastForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=np.zeros(n, dtype), Lf=Lf))
        s_ref = next(ref)

        def exchange(part, scal):
            # every scalar as a (hi, lo) pair of the working precision: hi = T(d), lo = T(d - hi) (col_pack_scalars_kernel)
            d = np.asarray(scal, np.float64)
            hi = d.astype(dtype)
            lo = (d - hi.astype(np.float64)).astype(dtype)
            slots = np.zeros(8 * world, dtype)
            slots[8 * rank:8 * rank + 8] = np.stack([hi, lo], axis=1).ravel()
            buf = torch.from_numpy(np.concatenate([part, slots]))
            pa.allreduce_sum_(buf)
            out = buf.numpy()
            sl = out[m:].astype(np.float64).reshape(world, 4, 2).sum(axis=2)
            return out[:m].copy(), (sl[:, 0].sum(), sl[:, 1].max(), sl[:, 2].sum(), sl[:, 3].sum())

        # init (state 1): x = x0, z = prox(x - gamma grad), z_prev = x   -- local slices
        x = np.zeros(ccnt, dtype)
        r, _ = exchange(A_loc @ x, np.zeros(4, dtype))
        r = r - b
        grad = A_loc.T @ r
        z, _ = o.NormL1(lam).prox(x - gamma * grad, gamma)
        z_prev = x.copy()
        seq = o.AdaptiveNesterovSequence(dtype(0))
        beta = seq.next(gamma)
        x_next = z + beta * (z - z_prev)
        r, _ = exchange(A_loc @ x_next, np.zeros(4, dtype))  # first half of iteration 2 on its own
        r = r - b
        for k in range(2, 40):
            s_ref = next(ref)
            x, z_prev = x_next, z
            beta2 = seq.next(gamma)
            grad = A_loc.T @ r  # local: the columns are ours
            z, _ = o.NormL1(lam).prox(x - gamma * grad, gamma)
            res = x - z
            x_next = z + beta2 * (z - z_prev)
            part, (gz, res_inf, dot_gr, res_sq) = exchange(
                A_loc @ x_next, np.array([np.sum(np.abs(z)), np.max(np.abs(res)), grad @ res, res @ res], dtype))
            r = part - b
            tol = 2e-4 if dtype == np.float32 else 1e-10
            assert np.max(np.abs(z - s_ref.z[coff:coff + ccnt])) <= tol * max(1.0, np.max(np.abs(s_ref.z))), k
            assert abs(res_inf - np.max(np.abs(s_ref.res))) <= tol * max(1.0, np.max(np.abs(s_ref.res))), k
            assert abs(lam * gz - s_ref.g_z) <= 10 * tol * max(1.0, abs(s_ref.g_z)), k
    # ---- row teams: the protocol of csrc/pg_gemv_tn4.hip (gemv_tnt_kernel<..., PEER>) on the CPU ----
    # every rank holds the row block A_p, b_p and the replicated n-vectors; per COLUMN the ranks exchange their partial dots
    # A_p[:, j]' r_p (here: one all-gather of the n partials instead of tagged granules) and every rank sums them IN RANK
    # ORDER -- the same bits on every rank; f = sum_p 1/2 ||A_p v - b_p||^2 is exchanged the same way.  No all-reduce in the
    # steady state; the iterates are the unsharded oracle's and identical on the two ranks, bit for bit.
    for dtype in (np.float32, np.float64):
        A, b, _ = o.synthetic_lasso(m, n, seed=0, dtype=dtype)
        lam = dtype(0.1) * dtype(np.max(np.abs(A.T @ b)))
        Lf = dtype(1.05 * np.linalg.norm(A.astype(np.float64), 2) ** 2)
        gamma = dtype(1) / Lf
        off, cnt = pa.shard_rows(m, world, rank)
        A_loc, b_loc = A[off:off + cnt], b[off:off + cnt]
        ref = iter(o.FastForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=np.zeros(n, dtype), Lf=Lf))
        next(ref)

        def gather_rank_order(vec):
            t = torch.from_numpy(np.ascontiguousarray(vec))
            parts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            tot = parts[0].numpy().copy()
            for q in range(1, world):
                tot = tot + parts[q].numpy()  # rank order, like the member-order sum of the granules
            return tot

        x = np.zeros(n, dtype)
        r = A_loc @ x - b_loc
        grad = gather_rank_order(A_loc.T @ r)
        z, _ = o.NormL1(lam).prox(x - gamma * grad, gamma)
        z_prev = x.copy()
        seq = o.AdaptiveNesterovSequence(dtype(0))
        x_next = z + seq.next(gamma) * (z - z_prev)
        r = A_loc @ x_next - b_loc
        for k in range(2, 40):
            s_ref = next(ref)
            x, z_prev = x_next, z
            beta2 = seq.next(gamma)
            grad = gather_rank_order(A_loc.T @ r)  # the per-column exchange inside the sweep
            z, _ = o.NormL1(lam).prox(x - gamma * grad, gamma)
            x_next = z + beta2 * (z - z_prev)
            r = A_loc @ x_next - b_loc  # the same sweep: A_p v from the column tiles
            f_next = float(gather_rank_order(np.array([0.5 * float(r.astype(np.float64) @ r.astype(np.float64))]))[0])
            tol = 2e-4 if dtype == np.float32 else 1e-10
            assert np.max(np.abs(z - s_ref.z)) <= tol * max(1.0, np.max(np.abs(s_ref.z))), k
            assert f_next >= 0
        both = gather_rank_order(np.concatenate([z.astype(np.float64), -z.astype(np.float64)]) if rank == 0 else
                                 np.concatenate([-z.astype(np.float64), z.astype(np.float64)]))
        assert np.all(both == 0), "the replicated iterate differs between the ranks"
    dist.barrier()
    if rank == 0:
        print("GLOO_SHARDED_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
