"""Hand-written edge-case scenarios (SURVEY.md §8c "G4") shared by the golden
generator and the parity tests.

A scenario is a starting position poked straight into the env state -- the way
the reference's own tests do it (``src/tests/test_mnk_integration.py:57-58,
123-124, 146-151``: ``env.boards[0, p, r, c] = 1`` without touching
``move_counts``) -- followed by a list of plies.  ``make_golden.py`` plays every
scenario on the imported reference and stores what came out; the tests replay it
on the oracle and on the HIP path and compare.

Fields: m, n, k; black / white = lists of (row, col) stones poked in;
side = side to move (poked into current_player), moves_made = poked move_counts;
plies = actions applied one after the other with ``env.step``.
"""

SCENARIOS = {
    # the reference's own test_env_mechanics_win: X X _ then (0,2)
    "row_win_3x3": dict(m=3, n=3, k=3, black=[(0, 0), (0, 1)], white=[], side=0, moves_made=0, plies=[2]),
    "col_win_3x3": dict(m=3, n=3, k=3, black=[(0, 1), (1, 1)], white=[], side=0, moves_made=0, plies=[7]),
    "diag_win_3x3": dict(m=3, n=3, k=3, black=[(0, 0), (1, 1)], white=[], side=0, moves_made=0, plies=[8]),
    "anti_win_3x3": dict(m=3, n=3, k=3, black=[(0, 2), (1, 1)], white=[], side=0, moves_made=0, plies=[6]),
    # white to move completes white's line
    "white_row_win": dict(m=3, n=3, k=3, black=[(0, 0)], white=[(1, 0), (1, 1)], side=1, moves_made=3, plies=[5]),
    # a line must not wrap around the board edge: (0,1),(0,2),(1,0) is not a row
    "no_wrap_row": dict(m=3, n=3, k=3, black=[(0, 1), (0, 2)], white=[], side=0, moves_made=0, plies=[3]),
    # ...nor along the anti-diagonal across the left edge, nor the diagonal across the right edge
    "no_wrap_anti": dict(m=4, n=4, k=3, black=[(0, 0), (0, 3)], white=[], side=0, moves_made=0, plies=[6]),
    "no_wrap_diag": dict(m=4, n=4, k=3, black=[(0, 3), (2, 0)], white=[], side=0, moves_made=0, plies=[13]),
    # a line that was already there counts even when the ply is elsewhere
    "old_line_counts": dict(m=3, n=3, k=3, black=[(2, 0), (2, 1), (2, 2)], white=[], side=0, moves_made=0, plies=[0]),
    # the other side's finished line is ignored when I move
    "their_line_ignored": dict(m=3, n=3, k=3, black=[], white=[(2, 0), (2, 1), (2, 2)], side=0, moves_made=0, plies=[0]),
    # last cell: win takes precedence over draw
    "win_beats_draw": dict(
        m=3, n=3, k=3,
        black=[(0, 0), (0, 1), (1, 2), (2, 0)], white=[(1, 0), (1, 1), (0, 2), (2, 1)], side=0, moves_made=8,
        plies=[8],
    ),
    # last cell, nobody wins: draw, reward 0
    "plain_draw": dict(
        m=3, n=3, k=3,
        black=[(0, 0), (0, 1), (1, 2), (2, 0)], white=[(0, 2), (1, 0), (1, 1), (2, 2)], side=0, moves_made=8,
        plies=[7],
    ),
    # occupied cell: both planes end up set, the move still counts, no error
    "overwrite_occupied": dict(m=3, n=3, k=3, black=[], white=[(1, 1)], side=0, moves_made=1, plies=[4]),
    # stepping a finished env keeps toggling the side and counting moves
    "keep_stepping_finished": dict(m=3, n=3, k=3, black=[(0, 0), (0, 1)], white=[], side=0, moves_made=0, plies=[2, 3, 4, 5]),
    # negative action wraps like torch indexing: -1 is the last cell
    "negative_action": dict(m=3, n=3, k=3, black=[], white=[], side=0, moves_made=0, plies=[-1, -9, -5]),
    # anti-diagonal on a non-square board
    "anti_4x6": dict(m=4, n=6, k=3, black=[(0, 5), (1, 4)], white=[], side=0, moves_made=0, plies=[15]),
    "diag_6x4": dict(m=6, n=4, k=4, black=[(2, 0), (3, 1), (4, 2)], white=[], side=0, moves_made=0, plies=[23]),
    # k = 5 on 9x9, all four directions, far from / touching the edges
    "row5_9x9": dict(m=9, n=9, k=5, black=[(8, 4), (8, 5), (8, 6), (8, 7)], white=[], side=0, moves_made=0, plies=[80]),
    "col5_9x9": dict(m=9, n=9, k=5, black=[(4, 8), (5, 8), (6, 8), (7, 8)], white=[], side=0, moves_made=0, plies=[80]),
    "diag5_9x9": dict(m=9, n=9, k=5, black=[(4, 4), (5, 5), (6, 6), (7, 7)], white=[], side=0, moves_made=0, plies=[80]),
    "anti5_9x9": dict(m=9, n=9, k=5, black=[(4, 4), (5, 3), (6, 2), (7, 1)], white=[], side=0, moves_made=0, plies=[72]),
    "four_is_not_five": dict(m=9, n=9, k=5, black=[(0, 0), (0, 1), (0, 2)], white=[], side=0, moves_made=0, plies=[3, 40, 5]),
    # six in a row also wins (>= k)
    "overline_9x9": dict(m=9, n=9, k=5, black=[(3, 0), (3, 1), (3, 2), (3, 4), (3, 5)], white=[], side=0, moves_made=0, plies=[30]),
    # 19x19: a line that straddles the u64 word boundaries of the packed layout
    "row5_19x19_word_edge": dict(m=19, n=19, k=5, black=[(3, 1), (3, 2), (3, 4), (3, 5)], white=[], side=0, moves_made=0, plies=[3 * 19 + 3]),
    "col5_19x19": dict(m=19, n=19, k=5, black=[(14, 18), (15, 18), (16, 18), (17, 18)], white=[], side=0, moves_made=0, plies=[18 * 19 + 18]),
    "diag5_19x19": dict(m=19, n=19, k=5, black=[(1, 1), (2, 2), (3, 3), (5, 5)], white=[], side=0, moves_made=0, plies=[4 * 19 + 4]),
    "anti5_19x19": dict(m=19, n=19, k=5, black=[(14, 4), (15, 3), (16, 2), (17, 1)], white=[], side=0, moves_made=0, plies=[18 * 19 + 0]),
    # 13x13 white wins on the diagonal
    "white_diag5_13x13": dict(m=13, n=13, k=5, black=[(0, 0)], white=[(8, 8), (9, 9), (10, 10), (11, 11)], side=1, moves_made=5, plies=[12 * 13 + 12]),
    # k = 1: the very first ply wins
    "k1_first_ply": dict(m=3, n=4, k=1, black=[], white=[], side=0, moves_made=0, plies=[5]),
}
