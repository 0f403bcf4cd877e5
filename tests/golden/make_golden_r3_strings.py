"""The synthetic canonical reaction strings of tests/golden/scoring_lists.npz (shared by make_golden_r3.py, which feeds them to the
reference, and the tests; no reference import here)."""


def scoring_strings():
    """3 conditions x 5 samples, condition-major.  True reactions: 'A.B>>P0', 'C>>P1', 'D.E.F>>P2'."""
    gen = ['B.A>>P0', 'A.B>>P0', 'A.X>>P0', 'A.B>>P0', 'Y>>P0',
           'Z>>P1', 'C>>P1', 'Z>>P1', '>>P1', 'C>>P1',
           'D.E>>P2', 'E.D>>P2', 'F.E.D>>P2', 'G>>P2', 'D.E>>P2']
    true = ['A.B>>P0'] * 5 + ['C>>P1'] * 5 + ['D.E.F>>P2'] * 5
    return gen, true
