"""Host logic of the device ingest that needs no GPU: how the chromosomes of a file are cut into inflate launches."""
from svision_amd.ingest_gpu import cut_groups


def test_groups_follow_the_limits_in_file_order():
    sizes = {t: 193 for t in range(20)}                       # the 20-window job: one window = 193 MB per chromosome
    groups = cut_groups(list(range(20)), sizes.__getitem__, [192, 600], 600)
    assert groups == [[0], [1, 2, 3], [4, 5, 6], [7, 8, 9], [10, 11, 12], [13, 14, 15], [16, 17, 18, 19]]
    assert [t for g in groups for t in g] == list(range(20))  # every chromosome once, in file order


def test_a_small_remainder_joins_the_group_in_front_and_a_large_one_does_not():
    sizes = {t: 193 for t in range(20)}
    assert cut_groups(list(range(20)), sizes.__getitem__, [192, 600], 600, merge_last=False)[-2:] == [[16, 17, 18], [19]]
    sizes[19] = 400                                           # more than half a group: launched on its own
    assert cut_groups(list(range(20)), sizes.__getitem__, [192, 600], 600)[-1] == [19]


def test_a_chromosome_larger_than_every_limit_is_a_group_of_its_own():
    sizes = {0: 3000, 1: 2900, 2: 100, 3: 100}
    assert cut_groups([0, 1, 2, 3], sizes.__getitem__, [192, 600], 600, merge_last=False) == [[0], [1], [2, 3]]
    assert cut_groups([0, 1, 2, 3], sizes.__getitem__, [192, 600], 600) == [[0], [1, 2, 3]]
    assert cut_groups([], sizes.__getitem__, [192, 600], 600) == []
    assert cut_groups([2], sizes.__getitem__, [192, 600], 600) == [[2]]
