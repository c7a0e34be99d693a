"""GPU parity of the dense pieces of the layer: tf.nn.dropout (gcn/layers.py:185) and
tf.matmul (gcn/layers.py:32-33) on the float32 MFMA pipe."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


def _t(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('shape', [(2708, 16), (3327, 7), (1, 1), (1001, 3)])      # 128-bit path and element path
@pytest.mark.parametrize('keep_prob', [0.5, 0.8, 1.0])
def test_dense_dropout_bit_exact(oracle, keep_prob, shape):
    from gcn_amd import sparse
    x = oracle.synthetic_dense(shape[0], shape[1], seed=31)
    ref, keep = oracle.dense_dropout(x, keep_prob, oracle.stream_key(99, 3))
    out = sparse.dropout(_t(x), keep_prob, 99, 3).cpu().numpy()
    assert np.array_equal(out, ref)
    # with a gate: the ReLU-backward form
    gate = oracle.synthetic_dense(shape[0], shape[1], seed=32)
    out_g = sparse.dropout(_t(x), keep_prob, 99, 3, gate=_t(gate)).cpu().numpy()
    assert np.array_equal(out_g, np.where(gate > 0, ref, np.float32(0)))
    # a row shard draws its slice of the unsplit matrix's mask, on either path
    half = shape[0] // 2
    if half:
        part = sparse.dropout(_t(x[half:]), keep_prob, 99, 3, index_offset=half * shape[1]).cpu().numpy()
        assert np.array_equal(part, ref[half:])


@pytest.mark.parametrize('m,n,k', [(2708, 7, 16), (3327, 6, 16), (19717, 3, 16), (1000, 16, 1433), (513, 130, 257),
                                   (4096, 512, 512), (1, 1, 1), (129, 33, 2), (300, 200, 17), (257, 64, 31),
                                   (256, 256, 64), (384, 96, 48),
                                   (25600, 512, 96), (128 * 300, 384, 32)])      # whole tiles, many of them
@pytest.mark.parametrize('trans_b', [False, True])
def test_gemm_mfma_against_float64(oracle, m, n, k, trans_b):
    from gcn_amd import sparse
    a = oracle.synthetic_dense(m, k, seed=41)
    b = oracle.synthetic_dense(k, n, seed=42)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    bt = _t(np.ascontiguousarray(b.T)) if trans_b else _t(b)
    out = sparse.gemm(_t(a), bt, trans_b=trans_b).cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(out - ref).max() <= REL_TOL * scale
    out_relu = sparse.gemm(_t(a), bt, trans_b=trans_b, relu=True).cpu().numpy()
    assert np.array_equal(out_relu, np.maximum(out, 0))


@pytest.mark.parametrize('m,n,k', [(128, 128, 32), (5000, 256, 64), (1000, 384, 96), (257, 128, 512), (1280, 512, 256),
                                   # 16 column tiles or more: the block order that walks groups of 8 panels (whole
                                   # groups, a last group of 3 rows of panels, fewer panels than one group)
                                   (128 * 19 + 7, 2048, 64), (128 * 64, 2304, 32), (300, 2048, 32)])
@pytest.mark.parametrize('trans_b', [False, True])
def test_whole_panel_form_gives_the_bits_of_the_guarded_form(oracle, m, n, k, trans_b):
    """Whole 128-row panels with n a multiple of 128 and k of 32 go through the kernel without guards
    (buffer addressing, scalar offsets), the rows left over and every other shape through the guarded one.
    Both add the products of an element in the same order: the same problem posed so that it takes the
    guarded kernel (four more columns) must give the same bits, with ReLU, with the gate of the ReLU /
    dropout backward, and on strided operands."""
    import torch
    from gcn_amd import sparse
    a = _t(oracle.synthetic_dense(m, k, seed=51))
    b_wide = oracle.synthetic_dense(k, n + 4, seed=52)
    if trans_b:
        bw = _t(np.ascontiguousarray(b_wide.T))            # (n + 4) x k
        b_part = bw[:n]
    else:
        bw = _t(b_wide)                                     # k x (n + 4)
        b_part = bw[:, :n]                                  # row pitch n + 4
    guarded = sparse.gemm(a, bw, trans_b=trans_b)[:, :n]
    panels = sparse.gemm(a, b_part, trans_b=trans_b)
    assert torch.equal(panels, guarded)
    ref = a.double() @ _t(b_wide[:, :n]).double()
    assert float((panels.double() - ref).abs().max()) <= REL_TOL * float(ref.abs().max())
    assert torch.equal(sparse.gemm(a, b_part, trans_b=trans_b, relu=True), torch.clamp(guarded, min=0))
    gate = _t(oracle.synthetic_dense(m, n, seed=53))
    gate = torch.where(gate > 0.3, gate, torch.zeros_like(gate))
    for keep in (0.5, 0.8, 0.25):                            # a power of two divides as a product with the inverse
        got = sparse.gemm_gated(a, b_part, gate, keep, trans_b=trans_b)
        want = torch.where(gate > 0, guarded / torch.full_like(guarded, keep), torch.zeros_like(guarded))
        assert torch.equal(got, want)
    # strided A, strided C and a strided gate
    a_wide = torch.zeros((m, k + 8), device='cuda')
    a_wide[:, 4:4 + k] = a
    c_wide = torch.full((m, n + 8), -3.0, device='cuda')
    sparse.gemm(a_wide[:, 4:4 + k], b_part, trans_b=trans_b, out=c_wide[:, 3:3 + n])
    assert torch.equal(c_wide[:, 3:3 + n], guarded)
    assert bool((c_wide[:, :3] == -3.0).all()) and bool((c_wide[:, 3 + n:] == -3.0).all())
    g_wide = torch.zeros((m, n + 4), device='cuda')
    g_wide[:, 1:1 + n] = gate
    sparse.gemm_gated(a_wide[:, 4:4 + k], b_part, g_wide[:, 1:1 + n], 0.5, trans_b=trans_b, out=c_wide[:, 3:3 + n])
    assert torch.equal(c_wide[:, 3:3 + n], torch.where(gate > 0, guarded * 2, torch.zeros_like(guarded)))


def test_relu_store_of_both_forms_on_nan_infinities_and_underflow():
    """The panel kernel's ReLU is one hardware maximum against 0, the guarded kernel's a compare and select.
    Rows 0..127 go through the first, rows 128..191 (duplicates of rows 0..63) through the second: they must
    hold the same bits where the accumulators are NaN (inf - inf, 0 * inf), +inf, -inf, a negative value that
    underflows to zero, and ordinary values of either sign."""
    import torch
    from gcn_amd import sparse
    m, n, k = 192, 128, 32
    a = torch.zeros((m, k), device='cuda')
    g = torch.Generator(device='cuda').manual_seed(4)
    a[:128] = torch.rand((128, k), device='cuda', generator=g) * 2 - 1
    a[0, 0], a[0, 1] = float('inf'), float('-inf')           # inf - inf against equal-sign columns of B -> NaN
    a[1, 0] = float('inf')                                     # +-inf, and 0 * inf = NaN where B is 0
    a[2, 0] = float('-inf')
    a[3] = 0.0
    a[3, 0] = -1e-30                                           # times 1e-30 in B: a negative product that underflows
    a[4] = 0.0                                                 # an all-zero row: accumulators stay +0
    a[128:] = a[:64]
    b = torch.rand((k, n), device='cuda', generator=g) * 2 - 1
    b[0, :8] = 0.0
    b[0, 8:16] = 1e-30
    out = sparse.gemm(a, b, relu=True)
    plain = sparse.gemm(a, b)
    assert bool(torch.isnan(plain[0]).any()) and bool(torch.isinf(plain[1]).any()) and bool(torch.isinf(plain[2]).any())
    assert torch.equal(out[:64].view(torch.int32), out[128:].view(torch.int32))
    assert torch.equal(plain[:64].view(torch.int32), plain[128:].view(torch.int32))
    # and both are `v > 0 ? v : 0` of the plain product, bit for bit (NaN and -0 become +0)
    want = torch.where(plain > 0, plain, torch.zeros_like(plain))
    assert torch.equal(out.view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize('pitch', [(1 << 22) - 64, (1 << 22) + 64])
def test_whole_panel_form_at_the_largest_row_pitch_it_takes(oracle, pitch):
    """The kernel without guards addresses a tile with 32-bit byte offsets: it takes operands whose 128 rows
    span less than 2^31 bytes (row pitch below 2^22 floats) and leaves wider ones to the guarded kernel.
    Just below the limit the offsets are the largest it ever forms; just above, the other kernel runs:
    the same bits as the contiguous problem either way, for A, for C and for the gate."""
    import torch
    from gcn_amd import sparse
    free, _ = torch.cuda.mem_get_info()
    if free < 24 * 2 ** 30:
        pytest.skip('needs 24 GiB of free HBM')
    m, n, k = 256, 128, 64
    a = _t(oracle.synthetic_dense(m, k, seed=54))
    b = _t(oracle.synthetic_dense(k, n, seed=55))
    want = sparse.gemm(a, b)
    big = torch.zeros((m, pitch), device='cuda')
    big[:, 8:8 + k] = a
    assert torch.equal(sparse.gemm(big[:, 8:8 + k], b), want)                  # pitch of A
    big.zero_()
    sparse.gemm(a, b, out=big[:, 4:4 + n])                                      # pitch of C
    assert torch.equal(big[:, 4:4 + n], want) and float(big[:, 4 + n:4 + n + 64].abs().max()) == 0.0
    gate = torch.where(want > 0.1, want, torch.zeros_like(want))
    gated = sparse.gemm_gated(a, b, gate, 0.5)
    gbig = torch.zeros((m, pitch), device='cuda')
    gbig[:, :n] = gate
    assert torch.equal(sparse.gemm_gated(a, b, gbig[:, :n], 0.5), gated)        # pitch of the gate
    # the transposed operand B [n x k] with that pitch
    del gbig
    big.zero_()
    big[:n, :k] = b.t()
    assert torch.equal(sparse.gemm(a, big[:n, :k], trans_b=True), want)
    del big
    torch.cuda.empty_cache()


def test_gemm_identity_with_asymmetric_operand():
    """A = I against an asymmetric B catches a swapped row/column map of the MFMA result."""
    import torch
    from gcn_amd import sparse
    n = 96
    b = torch.arange(n * 40, dtype=torch.float32, device='cuda').reshape(n, 40)
    out = sparse.gemm(torch.eye(n, device='cuda'), b)
    assert torch.equal(out, b)


@pytest.mark.parametrize('trans_b', [False, True])
def test_gemm_strided_and_unaligned_operands(oracle, trans_b):
    """Operands that are column slices of wider matrices: row strides larger than the width, a base
    address off the 16-byte grid (scalar loader) and on it (vector loader), output into a slice."""
    import torch
    from gcn_amd import sparse
    m, n, k = 260, 136, 72
    a = oracle.synthetic_dense(m, k, seed=51)
    b = oracle.synthetic_dense(k, n, seed=52)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    for shift in (1, 4):
        a_wide = torch.zeros((m, k + 12), dtype=torch.float32, device='cuda')
        a_wide[:, shift:shift + k] = _t(a)
        bsrc = np.ascontiguousarray(b.T) if trans_b else b
        b_wide = torch.zeros((bsrc.shape[0], bsrc.shape[1] + 8), dtype=torch.float32, device='cuda')
        b_wide[:, shift:shift + bsrc.shape[1]] = _t(bsrc)
        c_wide = torch.full((m, n + 8), -7.0, dtype=torch.float32, device='cuda')
        sparse.gemm(a_wide[:, shift:shift + k], b_wide[:, shift:shift + bsrc.shape[1]], trans_b=trans_b,
                    out=c_wide[:, 4:4 + n])
        got = c_wide.cpu().numpy()
        assert np.abs(got[:, 4:4 + n] - ref).max() <= REL_TOL * np.abs(ref).max()
        assert np.all(got[:, :4] == -7.0) and np.all(got[:, 4 + n:] == -7.0)      # nothing outside the slice


@pytest.mark.parametrize('m,n,k', [(2708, 7, 16), (2708, 16, 1433), (19717, 3, 16), (100000, 256, 256), (1000, 130, 33),
                                   (31, 5, 9), (1, 1, 1), (300000, 16, 200), (65, 128, 32), (70001, 384, 512)])
def test_weight_gradient_product_against_float64(oracle, m, n, k):
    """dW = H^T . dP (the reduction over the nodes), including shapes that split the rows over many
    blocks, ragged edges in every dimension and operands that are slices of wider matrices."""
    import torch
    from gcn_amd import sparse
    a = oracle.synthetic_dense(m, k, seed=61)
    g = oracle.synthetic_dense(m, n, seed=62)
    ref = a.astype(np.float64).T @ g.astype(np.float64)
    tol = REL_TOL * max(np.abs(ref).max(), np.sqrt(m))          # float32 sums of m products
    out = sparse.gemm_tn(_t(a), _t(g))
    assert np.abs(out.cpu().numpy() - ref).max() <= tol
    assert torch.equal(out, sparse.gemm_tn(_t(a), _t(g)))       # deterministic: fixed order of the partial sums
    a_wide = torch.zeros((m, k + 5), dtype=torch.float32, device='cuda')
    a_wide[:, 1:1 + k] = _t(a)
    c_wide = torch.full((k, n + 3), 3.0, dtype=torch.float32, device='cuda')
    sparse.gemm_tn(a_wide[:, 1:1 + k], _t(g), out=c_wide[:, 2:2 + n])
    got = c_wide.cpu().numpy()
    assert np.abs(got[:, 2:2 + n] - ref).max() <= tol
    # the unaligned operand takes the guarded loader, the aligned one (128 x 128 tiles) buffer loads with
    # scalar offsets: same row ranges, same order of additions, same bits
    assert torch.equal(c_wide[:, 2:2 + n], out)
    assert np.all(got[:, :2] == 3.0) and np.all(got[:, 2 + n:] == 3.0)


def test_gather_rows_of_a_column_range():
    """The send-buffer pack of the exchange: rows picked from a column slice of a wider matrix, with the
    128-bit path (aligned slice) and the element path (odd offset / width)."""
    import torch
    from gcn_amd import sparse
    gen = torch.Generator().manual_seed(3)
    src = torch.rand((5000, 72), generator=gen).cuda()
    rows = torch.randint(0, 5000, (12345,), generator=gen).cuda()
    for c0, c1 in ((0, 72), (8, 40), (3, 10), (64, 72)):
        out = torch.full((rows.numel(), c1 - c0 + 4), -1.0, device='cuda')
        sparse.gather_rows(src[:, c0:c1], rows, out[:, :c1 - c0])
        assert torch.equal(out[:, :c1 - c0], src[rows][:, c0:c1])
        assert bool((out[:, c1 - c0:] == -1.0).all())
    empty = torch.empty((0, 72), device='cuda')
    sparse.gather_rows(src, rows[:0], empty)


# (16 columns out of 4000 rows: from 4096 rows on the plain product takes the streaming kernel, which adds in another
# order than the tiled kernel the fused stores live in)
@pytest.mark.parametrize('m,n,k', [(5000, 256, 16), (1000, 130, 33), (129, 33, 2), (4000, 16, 256), (1, 1, 1)])
@pytest.mark.parametrize('trans_b', [False, True])
def test_gated_product_equals_product_then_gate(oracle, m, n, k, trans_b):
    """where(gate > 0, (A . B) / keep, 0) in the product's store: the bits of the product followed by the
    dropout / ReLU backward pass over it."""
    import torch
    from gcn_amd import sparse
    a = _t(oracle.synthetic_dense(m, k, seed=71))
    b = oracle.synthetic_dense(k, n, seed=72)
    bt = _t(np.ascontiguousarray(b.T)) if trans_b else _t(b)
    gate = _t(oracle.synthetic_dense(m, n, seed=73))
    gate = torch.where(gate > 0.3, gate, torch.zeros_like(gate))      # zeros and positives, as a dropped ReLU output
    plain = sparse.gemm(a, bt, trans_b=trans_b)
    for keep in (0.5, 0.8):
        got = sparse.gemm_gated(a, bt, gate, keep, trans_b=trans_b)
        want = torch.where(gate > 0, plain / torch.full_like(plain, keep), torch.zeros_like(plain))   # IEEE division
        assert torch.equal(got, want)
    wide = torch.full((m, n + 4), 2.0, device='cuda')
    sparse.gemm_gated(a, bt, gate, 0.5, trans_b=trans_b, out=wide[:, 2:2 + n])
    assert torch.equal(wide[:, 2:2 + n], torch.where(gate > 0, plain / torch.full_like(plain, 0.5), torch.zeros_like(plain)))
    assert bool((wide[:, :2] == 2.0).all()) and bool((wide[:, 2 + n:] == 2.0).all())


@pytest.mark.parametrize('m,n,k', [(5000, 256, 64), (1000, 130, 33), (129, 33, 2), (4000, 16, 256)])
@pytest.mark.parametrize('trans_b', [False, True])
def test_product_with_dropout_in_its_store(oracle, m, n, k, trans_b):
    """dropout(relu(A . B)) written by the GEMM: the bits of the product followed by the oracle's dense
    dropout with the same decision ids (a row block of a taller matrix, a column range of a wider one)."""
    import torch
    from gcn_amd import sparse
    a = _t(oracle.synthetic_dense(m, k, seed=81))
    b = oracle.synthetic_dense(k, n, seed=82)
    bt = _t(np.ascontiguousarray(b.T)) if trans_b else _t(b)
    plain = sparse.gemm(a, bt, trans_b=trans_b, relu=True)
    row0 = 333
    od = sparse.OutDropout(0.5, 17, 9, index_offset=row0 * n)
    fused = sparse.gemm(a, bt, trans_b=trans_b, relu=True, out_dropout=od)
    ref, _ = oracle.dense_dropout(plain.cpu().numpy(), 0.5, oracle.stream_key(17, 9), index_offset=row0 * n)
    assert np.array_equal(fused.cpu().numpy(), ref)
    wide = torch.full((m, n + 8), -2.0, device='cuda')
    sparse.gemm(a, bt, trans_b=trans_b, relu=True, out=wide[:, 4:4 + n],
                out_dropout=sparse.OutDropout(0.5, 17, 9, index_offset=row0 * n))
    assert torch.equal(wide[:, 4:4 + n], fused) and bool((wide[:, :4] == -2.0).all()) and bool((wide[:, 4 + n:] == -2.0).all())


def test_dropout_store_of_a_row_subset_draws_the_rows_own_decisions(oracle):
    """A GEMM over a subset of the rows of A, with the row map: every output row carries the decisions
    that row has in the product over all rows."""
    import torch
    from gcn_amd import sparse
    m, n, k = 3000, 256, 64
    a = _t(oracle.synthetic_dense(m, k, seed=91))
    b = _t(oracle.synthetic_dense(k, n, seed=92))
    full = sparse.gemm(a, b, relu=True, out_dropout=sparse.OutDropout(0.5, 5, 6))
    rows = torch.sort(torch.randperm(m, generator=torch.Generator().manual_seed(1))[:1111])[0].cuda()
    part = sparse.gemm(a[rows].contiguous(), b, relu=True, out_dropout=sparse.OutDropout(0.5, 5, 6, pitch=n, row_ids=rows))
    assert torch.equal(part, full[rows])


@pytest.mark.parametrize('m,n,k', [(640, 128, 64), (1000, 256, 96), (4096, 256, 256), (77, 16, 40), (300, 7, 16),
                                    (513, 130, 33)])
@pytest.mark.parametrize('keep', [0.5, 0.8])
def test_dropout_in_the_left_operand_load_equals_the_separate_pass(oracle, m, n, k, keep):
    """gcn_dropout_bits + gcn_gemm_masked_f32 / gcn_gemm_tn_masked_f32: the dropout of a layer input applied as
    the product loads it.  The bits are the oracle's decisions; both products give the bits of the same
    product on the matrix a separate dropout pass writes (whole panels, ragged shapes, row blocks)."""
    import torch
    from gcn_amd import sparse
    seed, stream, first_row = 21, 5, 1000
    a = _t(oracle.synthetic_dense(m, k, seed=61))
    b = _t(oracle.synthetic_dense(k, n, seed=62))
    g = _t(oracle.synthetic_dense(m, n, seed=63))
    bits = sparse.dropout_bits(m, k, keep, seed, stream, index_offset=first_row * k)
    dropped = sparse.dropout(a, keep, seed, stream, index_offset=first_row * k)
    _, kept = oracle.dense_dropout(a.cpu().numpy(), np.float32(keep), oracle.stream_key(seed, stream),
                                   index_offset=first_row * k)
    words = bits.cpu().numpy().view(np.uint32)
    unpacked = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(m, -1)[:, :k].astype(bool)
    assert np.array_equal(unpacked, kept)
    assert np.array_equal(unpacked, dropped.cpu().numpy() != 0) or (a == 0).any()
    assert torch.equal(sparse.gemm_masked(a, bits, keep, b), sparse.gemm(dropped, b))
    assert torch.equal(sparse.gemm_masked(a, bits, keep, b, relu=True), sparse.gemm(dropped, b, relu=True))
    assert torch.equal(sparse.gemm_masked(a, bits, keep, b.t().contiguous(), trans_b=True), sparse.gemm(dropped, b))
    assert torch.equal(sparse.gemm_tn_masked(a, bits, keep, g), sparse.gemm_tn(dropped, g))
    # strided operand and mask rows wider than needed
    a_wide = torch.zeros((m, k + 8), device='cuda')
    a_wide[:, 4:4 + k] = a
    bits_wide = torch.zeros((m, bits.shape[1] + 3), dtype=torch.int32, device='cuda')
    bits_wide[:, :bits.shape[1]] = bits
    assert torch.equal(sparse.gemm_masked(a_wide[:, 4:4 + k], bits_wide[:, :bits.shape[1]], keep, b),
                       sparse.gemm(dropped, b))
    assert torch.equal(sparse.gemm_tn_masked(a_wide[:, 4:4 + k], bits_wide[:, :bits.shape[1]], keep, g),
                       sparse.gemm_tn(dropped, g))


@pytest.mark.parametrize('m,n,k', [(1280, 256, 256), (640, 128, 64), (1000, 384, 96)])
@pytest.mark.parametrize('trans_b', [False, True])
def test_dropout_store_of_whole_panels_equals_the_guarded_form(oracle, m, n, k, trans_b):
    """The store with dropout on whole 128-row panels (one hash per element in the unguarded kernel's
    epilogue) against the same problem posed so that it takes the guarded kernel (four more columns), and
    against the product followed by a separate dropout pass; with row_ids and with a row block's offset."""
    import torch
    from gcn_amd import sparse
    a = _t(oracle.synthetic_dense(m, k, seed=71))
    b_wide = oracle.synthetic_dense(k, n + 4, seed=72)
    if trans_b:
        bw = _t(np.ascontiguousarray(b_wide.T))
        b_part = bw[:n]
    else:
        bw = _t(b_wide)
        b_part = bw[:, :n]
    for keep in (0.5, 0.8):
        drop = sparse.OutDropout(keep, 31, 7, index_offset=5 * (n + 4), pitch=n + 4)
        guarded = sparse.gemm(a, bw, trans_b=trans_b, relu=True, out_dropout=drop)[:, :n]
        panels = sparse.gemm(a, b_part, trans_b=trans_b, relu=True, out_dropout=drop)
        assert torch.equal(panels, guarded)
        plain = sparse.gemm(a, b_part, trans_b=trans_b, relu=True)
        sep = sparse.dropout(plain, keep, 31, 7)
        assert torch.equal(sparse.gemm(a, b_part, trans_b=trans_b, relu=True, out_dropout=sparse.OutDropout(keep, 31, 7)), sep)
        ids = torch.randperm(3 * m, device='cuda')[:m].contiguous()
        by_rows = sparse.OutDropout(keep, 31, 7, pitch=n + 4, row_ids=ids)
        assert torch.equal(sparse.gemm(a, b_part, trans_b=trans_b, out_dropout=by_rows),
                           sparse.gemm(a, bw, trans_b=trans_b, out_dropout=by_rows)[:, :n])


@pytest.mark.parametrize('m,n,k', [(4096, 16, 256), (5000, 16, 16), (70001, 7, 48), (4100, 1, 64), (9000, 17, 32),
                                   (4097, 32, 448), (6000, 16, 768), (12345, 24, 256)])
def test_streaming_form_of_the_product_with_few_output_columns(oracle, m, n, k):
    """n <= 32 with k in whole 16-k steps and at least 4096 rows: the kernel that feeds A from global memory
    straight into 16 x 16 x 4 MFMAs (hidden -> classes on a large graph).  Against float64, with ReLU, on
    strided operands and a strided output whose neighbours must stay untouched, and next to the tiled kernel
    (the same product posed with one row fewer than the streaming form takes... is not possible: the tiled
    kernel is reached through a k that is not a multiple of 16)."""
    import torch
    from gcn_amd import sparse
    a = _t(oracle.synthetic_dense(m, k, seed=61))
    b = _t(oracle.synthetic_dense(k, n, seed=62))
    ref = a.double() @ b.double()
    scale = float(ref.abs().max())
    got = sparse.gemm(a, b)
    assert float((got.double() - ref).abs().max()) <= REL_TOL * scale
    assert torch.equal(sparse.gemm(a, b), got)                      # deterministic
    assert torch.equal(sparse.gemm(a, b, relu=True), torch.clamp(got, min=0))
    # the tiled kernel on the same numbers: k + 4 with four zero columns / rows appended
    a_pad = torch.zeros((m, k + 4), device='cuda')
    a_pad[:, :k] = a
    b_pad = torch.zeros((k + 4, n), device='cuda')
    b_pad[:k] = b
    tiled = sparse.gemm(a_pad, b_pad)
    assert float((got - tiled).abs().max()) <= 4e-6 * scale
    # strided A (row pitch k + 8, start 4 floats in), strided B, strided C
    a_wide = torch.zeros((m, k + 8), device='cuda')
    a_wide[:, 4:4 + k] = a
    b_wide = torch.zeros((k, n + 3), device='cuda')
    b_wide[:, 1:1 + n] = b
    c_wide = torch.full((m, n + 5), -3.0, device='cuda')
    sparse.gemm(a_wide[:, 4:4 + k], b_wide[:, 1:1 + n], out=c_wide[:, 2:2 + n])
    assert torch.equal(c_wide[:, 2:2 + n], got)
    assert bool((c_wide[:, :2] == -3.0).all()) and bool((c_wide[:, 2 + n:] == -3.0).all())
