"""Host logic of the device ingest that needs no GPU: how the chromosomes of a file are cut into inflate launches."""
from svision_amd.ingest_gpu import cut_groups


def test_groups_follow_the_limits_in_file_order():
    sizes = {t: 193 for t in range(20)}                       # the 20-window job: one window = 193 MB per chromosome
    groups = cut_groups(list(range(20)), sizes.__getitem__, [192, 600], 600)
    # (round 6: the remainder, a third of a group, is a launch of its own -- it joined the group in front of it until then)
    assert groups == [[0], [1, 2, 3], [4, 5, 6], [7, 8, 9], [10, 11, 12], [13, 14, 15], [16, 17, 18], [19]]
    # the shipped limits (second group 400 MB): [1, 2, 3, 3, 3, 3, 3, 2]
    assert [len(g) for g in cut_groups(list(range(20)), sizes.__getitem__, [192, 400], 600)] == [1, 2, 3, 3, 3, 3, 3, 2]
    assert [t for g in groups for t in g] == list(range(20))  # every chromosome once, in file order


def test_a_small_remainder_joins_the_group_in_front_and_a_large_one_does_not():
    sizes = {t: 193 for t in range(20)}
    assert cut_groups(list(range(20)), sizes.__getitem__, [192, 600], 600, merge_last=False)[-2:] == [[16, 17, 18], [19]]
    sizes[19] = 140                                           # at most a quarter of a group: joins the group in front of it
    assert cut_groups(list(range(20)), sizes.__getitem__, [192, 600], 600)[-1] == [16, 17, 18, 19]
    sizes[19] = 193                                           # more than a quarter: launched on its own (profiles/r06_group_sweep.txt)
    assert cut_groups(list(range(20)), sizes.__getitem__, [192, 600], 600)[-1] == [19]


def test_a_chromosome_larger_than_every_limit_is_a_group_of_its_own():
    sizes = {0: 3000, 1: 2900, 2: 100, 3: 100}
    assert cut_groups([0, 1, 2, 3], sizes.__getitem__, [192, 600], 600, merge_last=False) == [[0], [1], [2, 3]]
    assert cut_groups([0, 1, 2, 3], sizes.__getitem__, [192, 600], 600) == [[0], [1], [2, 3]]          # 200 MB: more than a quarter of a group
    sizes[2] = sizes[3] = 70
    assert cut_groups([0, 1, 2, 3], sizes.__getitem__, [192, 600], 600) == [[0], [1, 2, 3]]
    assert cut_groups([], sizes.__getitem__, [192, 600], 600) == []
    assert cut_groups([2], sizes.__getitem__, [192, 600], 600) == [[2]]
