"""Oracle (test infrastructure): OctoMap insertion + binary (.bt) export restated.  PARITY UNPINNED.

The reference (octomap/txt_transfer_octomap.py:16-36, octomap/ply_transfer_octomap.py:16-48) calls the
third-party OctoMap library through its python binding: `OcTree(0.1)`, `updateNode(xyz, True)` per point,
`updateInnerOccupancy()`, `writeBinary(path)`.  The binding is NOT pinned in requirements.txt and is absent
here, so nothing can be run to pin this restatement; it follows OctoMap's published semantics
(octomap 1.9.x, OcTreeBaseImpl / OcTreeKey / AbstractOccupancyOcTree / OcTree::writeBinaryNode):

  * point3d holds FLOATS; key per axis = (int)floor((1/res) * (double)(float)coord) + 32768, tree depth 16,
    a point is ignored unless all three keys are in [0, 65535];
  * every visited leaf has positive log-odds after a hit, so writeBinary's toMaxLikelihood() makes each an
    occupied leaf: the .bt depends only on the SET of hit voxels;
  * prune(): a node whose 8 children exist, are leaves and are equal collapses into a leaf, bottom-up;
  * child index = xbit | ybit<<1 | zbit<<2 at each level (MSB first);
  * node record: 2 bits per child (00 unknown, 01->bit(2i+1) occupied leaf, 11 inner), children 0-3 in the
    first byte, 4-7 in the second, then the inner children's records depth-first in child order;
  * header: "# Octomap OcTree binary file\n# (feel free to add / change comments, but leave the first line as
    it is!)\n#\nid OcTree\nsize <nodes>\nres <res>\ndata\n".
"""
import numpy as np

TREE_DEPTH = 16
TREE_MAX_VAL = 32768


def voxel_keys(xyz, res=0.1):
    """[N,3] int64 keys and a validity mask, OcTreeBaseImpl::coordToKeyChecked on float coordinates."""
    p = np.asarray(xyz, dtype=np.float32).astype(np.float64)
    factor = 1.0 / res
    with np.errstate(invalid="ignore"):
        k = np.floor(factor * p)
    ok = np.isfinite(k).all(axis=1)
    k = np.where(np.isfinite(k), k, 0).astype(np.int64) + TREE_MAX_VAL
    ok &= ((k >= 0) & (k < 2 * TREE_MAX_VAL)).all(axis=1)
    return k, ok


def morton(keys):
    """48-bit code, 3 bits per level, level 15 (MSB of the keys) first; x is the low bit of each triple."""
    keys = np.asarray(keys, dtype=np.uint64)
    m = np.zeros(keys.shape[0], dtype=np.uint64)
    for b in range(TREE_DEPTH):
        for axis in range(3):
            m |= ((keys[:, axis] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + axis)
    return m


def occupied_set(xyz, res=0.1):
    """Sorted unique Morton codes of the voxels hit by the cloud, and the number of ignored points."""
    k, ok = voxel_keys(xyz, res)
    return np.unique(morton(k[ok])), int((~ok).sum())


def write_bt_bytes(codes, res=0.1):
    """The .bt file bytes for a sorted array of unique Morton codes (the occupied leaves at depth 16)."""
    codes = np.asarray(codes, dtype=np.uint64)
    body = bytearray()
    n_nodes = 0

    def full(count, child_depth):
        return count == 8 ** (TREE_DEPTH - child_depth)

    def node(lo, hi, depth):
        # node at `depth` (root = 0) owning codes[lo:hi]; it is an INNER node by construction
        nonlocal n_nodes
        n_nodes += 1
        shift = np.uint64(3 * (TREE_DEPTH - 1 - depth))
        sub = (codes[lo:hi] >> shift) & np.uint64(7)
        bounds = lo + np.searchsorted(sub, np.arange(9, dtype=np.uint64))
        b = [0, 0]
        inner = []
        for c in range(8):
            clo, chi = int(bounds[c]), int(bounds[c + 1])
            if chi == clo:
                continue
            if depth + 1 == TREE_DEPTH or full(chi - clo, depth + 1):
                b[c // 4] |= 2 << (2 * (c % 4))        # occupied leaf (possibly a pruned subtree)
                n_nodes += 1
            else:
                b[c // 4] |= 3 << (2 * (c % 4))
                inner.append((clo, chi))
        body.extend(bytes(b))
        for clo, chi in inner:
            node(clo, chi, depth + 1)

    if codes.shape[0]:
        if full(codes.shape[0], 0):
            n_nodes = 1                                  # the whole universe pruned into the root: no record
            body.extend(b"\x00\x00")
        else:
            node(0, codes.shape[0], 0)
    head = ("# Octomap OcTree binary file\n# (feel free to add / change comments, but leave the first line as it is!)\n#\n"
            "id OcTree\nsize %d\nres %s\ndata\n" % (n_nodes, format_res(res)))
    return head.encode() + bytes(body), n_nodes


def format_res(res):
    """std::ostream << double with default precision 6 (%g)."""
    return "%g" % res


def read_bt_leaves(data):
    """Parse .bt bytes back to (res, sorted array of (morton_prefix, depth) occupied leaves) -- round-trip check."""
    head_end = data.index(b"data\n") + 5
    header = data[:head_end].decode().split("\n")
    res = float([h for h in header if h.startswith("res ")][0].split()[1])
    size = int([h for h in header if h.startswith("size ")][0].split()[1])
    body = data[head_end:]
    pos = 0
    leaves = []
    count = 0

    def node(prefix, depth):
        nonlocal pos, count
        count += 1
        b0, b1 = body[pos], body[pos + 1]
        pos += 2
        inner = []
        for c in range(8):
            bits = ((b0 if c < 4 else b1) >> (2 * (c % 4))) & 3
            if bits == 2:
                leaves.append(((prefix << 3) | c, depth + 1))
                count += 1
            elif bits == 3:
                inner.append(c)
            elif bits == 1:
                raise ValueError("free leaf in a hits-only tree")
        for c in inner:
            node((prefix << 3) | c, depth + 1)

    if size:
        node(0, 0)
    return res, size, count, leaves
