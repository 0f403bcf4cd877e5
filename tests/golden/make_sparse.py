"""PyG-style sparse batch (x, edge_index, edge_attr, batch, ...) from a dense synthetic batch: test input builder shared by the
golden generator's to_dense case and the GPU tests (no reference code in here)."""
import torch
import torch.nn.functional as F


def sparse_from_dense(d, self_loops=False, seed=0):
    """Real nodes only, every non-'none' ordered pair one edge, edges in random order (PyG fixes none)."""
    ke = d['E'].shape[-1]
    g = torch.Generator().manual_seed(seed)
    xs, batch, ams, mols, ei, ea = [], [], [], [], [], []
    off = 0
    for b in range(d['X'].shape[0]):
        nb = int(d['node_mask'][b].sum())
        xs.append(d['X'][b, :nb]); batch.append(torch.full((nb,), b)); ams.append(d['atom_map_numbers'][b, :nb]); mols.append(d['mol_assignments'][b, :nb])
        cls = d['E'][b, :nb, :nb].argmax(-1)
        i, j = (cls > 0).nonzero(as_tuple=True)
        perm = torch.randperm(i.numel(), generator=g)
        i, j = i[perm], j[perm]
        ei.append(torch.stack([i + off, j + off])); ea.append(F.one_hot(cls[i, j], ke).float())
        if self_loops:                                             # self loops must be dropped by to_dense
            ei.append(torch.tensor([[off], [off]])); ea.append(F.one_hot(torch.tensor([2]), ke).float())
        off += nb
    return dict(x=torch.cat(xs), batch=torch.cat(batch), edge_index=torch.cat(ei, 1), edge_attr=torch.cat(ea),
                y=torch.zeros(d['X'].shape[0], 1), mask_atom_mapping=torch.cat(ams), mol_assignment=torch.cat(mols))
