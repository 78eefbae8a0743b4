// cc_policy.h — the window policy of the exact online phase as a pure host object.
//
// Between two batches of windows the host reads the device's control block once and decides how the next batch runs:
// window size, validation rounds enqueued per window, windows per batch, lookahead scans on / off, dirty scans launched
// or not, pruned or plain snapshot scans, scans split over the ranks of a group or not.  None of this can change a
// result (every combination is exact), but inside a multi-GPU group every rank has to take the SAME decisions - they
// determine the sequence of collectives -, so the policy must be a function of the device counters alone, which are
// identical on all ranks.  That is what this file isolates: WindowPolicy sees nothing but PolicyConfig (the call's
// constants), PolicyCarry (three integers a handle keeps from call to call) and one BatchObs per batch (cumulative
// counters as read back); no clock, no pointer, no HIP.  cc_policy_replay (C-ABI) runs it over recorded observations on
// a machine without a GPU (tests/test_window_policy.py).
// The one wall-clock rule of the library - handing a stream of short, truncated windows to the sequential kernel when
// that measures faster - stays in cc_api.hip, is switched off inside a group, and only consumes Decision::bad from here.
#pragma once
#include <algorithm>
#include <cstdint>
#include <limits>

#include "../../include/chronoclust_hip.h"

namespace cc {

// (two thresholds since round 6: a pruned chain spends a third of a plain scan's time on the same rows, so it pays to split it
// only from a table about three times as large; shard_min_row_dims_pruned == 0 - configurations recorded before - means "the
// same as for plain scans")
inline bool policy_want_shard(const cc_policy_config& c, int m_rows, bool pruned)
{
    const long long thr = (pruned && c.shard_min_row_dims_pruned > 0) ? c.shard_min_row_dims_pruned : c.shard_min_row_dims;
    return c.can_shard != 0 && (long long)m_rows * c.d >= thr;
}

// Points per millisecond the sequential kernel that WOULD take over is assumed to manage before it has been measured in
// this call (the takeover rule of cc_api.hip compares the windows' measured rate with it).  k_seq_r (rows in registers,
// d <= 4) ~0.6 us per point, k_seq (table in LDS) ~1.3 us, whatever the data.  k_seq_g (table in HBM, from `seq_cap` rows
// on) is one workgroup of 1 024 threads that walks rows / 1 024 rows per thread: 8-14 us per point measured at 150-450
// rows (profiles/r05_tool_seq_g.txt) - 100 points per ms up to 1 024 rows, and in proportion to 1 024 / rows beyond: at
// 20 000-50 000 rows a truncating stream whose windows still make 30-140 points per ms must not be handed a 32 768-point
// stint of a kernel that manages 2-5.
inline double seq_rate_guess(int d, int m_rows, int seq_cap, bool allow_seq_r, bool allow_seq_g)
{
    if (allow_seq_g && m_rows >= seq_cap) return 100.0 * std::min(1.0, 1024.0 / (double)std::max(1, m_rows));
    if (allow_seq_r && d >= 2 && d <= 4) return 1500.0;
    return 700.0;
}

class WindowPolicy {
public:
    static constexpr int kStartSmall = 256;   // window on a (nearly) empty table
    static constexpr int kFastBelow = 4096;   // below this size a clean batch quadruples the window, batches are two windows

    WindowPolicy(const cc_policy_config& cfg, const cc_policy_carry& carry) : c_(cfg), k_(carry) {}

    // the first batch of a call that starts at `cursor` with `m_rows` table rows
    cc_policy_decision start(long long cursor, int m_rows)
    {
        const int early0 = c_.early_window > 0 ? c_.early_window : 4096;
        // start where the previous call settled; a (nearly) empty table starts small and grows by doubling (a new
        // timepoint begins with whatever changed since the last one - decayed weights, new populations -, which takes a
        // few validation rounds per window: not with the largest window the previous one ended on)
        if (k_.adapt_win > 0) win_cfg_ = std::min(c_.window, c_.resume ? k_.adapt_win : std::min(k_.adapt_win, early0));
        else win_cfg_ = std::min(c_.window, (m_rows < 1024) ? kStartSmall : early0);
        rcur_ = c_.rounds_max;
        batch_windows_ = (win_cfg_ < 1024) ? 2 : std::max(2, c_.windows_per_sync / 4);
        la_on_ = c_.lookahead == 3;
        nodirty_ = false;
        sparse_ = false;
        prune_on_ = c_.prune_applicable != 0 && c_.prune_mode != 0;  // (independent of the split: k_scan_p takes a row range)
        shard_on_ = policy_want_shard(c_, m_rows, prune_on_);
        prune_resume_at_ = 0;
        prune_backoff_ = 65536;
        guess_on_ = false;
        lean_on_ = false;
        clean_guess_batches_ = 0;
        guess_resume_at_ = 0;
        stalled_ = 0;
        first_batch_ = true;
        prev_ = cc_policy_obs{};
        prev_.cursor = cursor;
        prev_.m_rows = m_rows;
        probe_ok_ = false;
        cc_policy_decision d0 = decision(0, 0);
        prev_probe_ = d0.probe != 0;
        return d0;
    }

    // the stream comes back from the sequential kernel: a fresh window at the cursor, dirty scans launched again, no
    // lookahead scan pending; the counters of the windows were not touched meanwhile
    cc_policy_decision after_sequential(long long cursor, int m_rows)
    {
        prev_.cursor = cursor;
        prev_.m_rows = m_rows;
        nodirty_ = false;
        sparse_ = false;
        la_on_ = c_.lookahead == 3;
        return decision(1, 0);
    }

    // a batch of windows has run; `o` = the cumulative counters now
    cc_policy_decision after_batch(const cc_policy_obs& o)
    {
        // (the configured window while a microcluster takes at most a dozen of a window's points on average; with fewer
        // microclusters - the C4 shape: 2 000 - a window of 49 152 points makes chains of more claimants than the member lists hold
        // the rule rather than the exception: 32 768 there, as before round 6)
        const int Rmax = c_.rounds_max;
        const int win = (c_.window > 32768 && (long long)o.m_rows * 12 < (long long)c_.window) ? 32768 : c_.window;
        const int early_win = c_.early_window > 0 ? c_.early_window : 4096;
        const long long done_before = prev_.cursor, done = o.cursor;
        // ---- pruned scans: on while the rows they still evaluate in full stay a minority ----
        const unsigned long long pr = o.prune_rows - prev_.prune_rows, pfu = o.prune_full - prev_.prune_full;
        if (prune_on_) {
            // More than one (wave, row) pair in twelve evaluated in full: the prefix scan and the tests cost more than
            // they save (start-up: the points' own microclusters do not exist yet; overlapping data).  A completed row
            // costs about ten times a row of the plain scan - its operands arrive by scalar loads nothing hides -, so the
            // break-even share is 15 - 25 % (d = 20 .. 40), not one half.
            // The plain scan then runs for a stretch of points before the next try: 65 536 after the first failed try,
            // twice as many after each further one in a row (a stream that is all start-up - forty points per
            // microcluster - should not pay for a try every few batches), back to 65 536 after a try that paid.
            if (pr > 0 && pfu * 12 > pr) {
                prune_resume_at_ = o.cursor + prune_backoff_;
                prune_backoff_ = std::min<long long>(1ll << 20, prune_backoff_ * 2);
                probe_ok_ = false;
            } else if (pr > 0) prune_backoff_ = 65536;
            // A batch that commits nothing although dirty scans were launched.  With guessed thresholds that is legitimate:
            // a lean scan missed the batch's first point, the list of missed points was cut there (CC_MISSED_CAP), or a
            // point's outlier list started with a bound (k_decide, stage 1) - the seeded chain comes back for 2^18 points.
            // With SEEDED thresholds first candidates are exact and the first point of a window is always decidable:
            // should such a batch still commit nothing, the plain scan takes over for good.
            // The SECOND batch in a row without progress - whatever the first one was put down to (guessed thresholds, dirty
            // scans that were not launched) - ends the pruned scans for the call: plain scans, dirty scans launched and no
            // lookahead always decide a window's first point.
            if (done == done_before && stalled_ >= 1) prune_resume_at_ = std::numeric_limits<long long>::max();
            else if (done == done_before && !nodirty_) {
                if (guess_on_) guess_resume_at_ = o.cursor + (1ll << 18);
                else prune_resume_at_ = std::numeric_limits<long long>::max();
            }
        }
        // whatever the cause, a call must not spin: a batch without progress is legitimate (a point refused for want of the
        // dirty scans or of the seeded chain idles the rest of its batch; each cause once: guessed thresholds -> seeded ->
        // plain scans, dirty scans not launched -> launched), not five times in a row
        stalled_ = (done == done_before) ? stalled_ + 1 : 0;
        // ---- validation rounds enqueued per window: what the last batch needed ----
        int used = 1;
        for (int r = 1; r <= CC_POLICY_MAX_ROUNDS; ++r)
            if (o.round_hist[r] - prev_.round_hist[r] > 0) used = r;
        const long long trunc_batch = o.stat_truncated - prev_.stat_truncated;
        const int rounds_batch = rcur_;
        const long long unk_batch = o.stat_trunc_unknown - prev_.stat_trunc_unknown;
        // (a window that stopped at a point whose decision could not be made - both snapshot candidates changed and no live
        // version beats the bound - would stop there after any number of rounds: only windows whose decisions were still
        // moving ask for one more)
        const long long trunc_moving = trunc_batch - unk_batch;
        if (trunc_batch > 0 && nodirty_) {
            // points refused for want of the dirty scans: the next batch launches them again, nothing else changes
        } else if (trunc_moving > 0) rcur_ = std::min(Rmax, std::max(used, rcur_) + 1);
        else rcur_ = std::max(1, std::min(rcur_, used));
        // ---- window size ----
        //  - While many MCs are being created the validation of a window is quadratic in its size (their versions cannot
        //    be pruned): at most `early_win` there, the configured size once the table is stable.
        //  - When windows commit short of their size (few MCs, overlapping data: the validation frontier stops early) the
        //    speculated remainder is wasted: aim at the average committed length; grow back by doubling while nothing
        //    is truncated.
        const long long grew = (long long)o.m_rows - prev_.m_rows, pts = o.cursor - prev_.cursor;
        // while MCs are being created a window may need one round more than the last batch's windows did (a promotion inside
        // the window, a link of round 0 that did not hold - cc_link.h): one spare round.  (Before round 0 linked a window's
        // creators such windows took three rounds and got all the rounds there are; `used` says so itself when they do.)
        if (pts > 0 && grew * 50 > pts) rcur_ = std::min(Rmax, std::max(rcur_, used + 1));
        const long long wins = o.stat_windows - prev_.stat_windows;
        //  - The same holds while MCs are being promoted: a promoted MC competes in a list it was not scanned for, so
        //    the dirty scans run unpruned; the device counts the point tiles whose dirty scan ran.
        const long long tiles = o.stat_tiles - prev_.stat_tiles, dtiles = o.stat_dirty_tiles - prev_.stat_dirty_tiles;
        //    What counts is how many POINTS needed rows that only a dirty scan covers, not how many tiles held such a
        //    point: a few per cent of the points (the first points of a new population; points between two far pcore MCs
        //    while their own MC is still an outlier MC) put one into most tiles.  While they stay under a sixteenth of
        //    the batch's points their dirty scans run point by point (the sparse dirty scans: compacted tiles of just
        //    those points), the tiles' scans are not launched and the windows run at full size with lookahead.
        //    (c_.allow_sparse is the divisor: one point in allow_sparse.  Every such point costs a pass over the window's
        //    live version rows, so the share has to be small: the work grows with the square of the window.)
        const long long flagged = o.stat_unsafe - prev_.stat_unsafe;
        const bool few_flagged = c_.allow_sparse > 0 && pts > 0 && flagged * c_.allow_sparse <= pts;
        const bool unpruned = tiles > 0 && dtiles * 2 > tiles && !few_flagged;
        const int target = ((pts > 0 && grew * 50 > pts) || unpruned) ? std::min(win, early_win) : win;
        int want = win_cfg_;
        if (nodirty_ && trunc_batch > 0) {
            // (see above: windows stopped at points that needed the dirty scans)
        } else if (trunc_batch * 4 >= wins && trunc_batch > 0 && rounds_batch < Rmax && unk_batch * 2 < trunc_batch) {
            // windows stopped short because their decisions were still moving, with fewer validation rounds enqueued
            // than allowed: more rounds (above) are the remedy, not a shorter window
            k_.clean_batches = 0;
        } else if (trunc_batch * 4 >= wins && trunc_batch > 0) {
            // a quarter or more of the windows stopped short: the window is too long for this data
            const long long avg = pts / wins;
            want = (int)std::min<long long>(target, std::max<long long>(128, ((avg + 63) / 64) * 64));
            k_.clean_batches = 0;
            k_.since_shrink = 0;
        } else {
            // an occasional short window (one more validation round needed than enqueued) is no reason to shrink; grow
            // by doubling after one clean batch, after two if a shrink is recent
            ++k_.since_shrink;
            if (trunc_batch == 0) ++k_.clean_batches;
            const int need = (k_.since_shrink > 8 || want < kFastBelow) ? 1 : 2;
            if (trunc_batch == 0 && k_.clean_batches >= need) {
                // a batch in which no window was cut short and no point tile needed its dirty scan (the table has
                // settled: nothing created or promoted any more) goes straight to the full size
                const bool settled = tiles > 0 && (dtiles == 0 || few_flagged) && grew == 0;
                want = settled ? target : std::min(target, std::max(want, 64) * (want < kFastBelow ? 4 : 2));
            }
        }
        want = std::min(want, target);
        k_.adapt_win = want;
        // ---- dirty scans, lookahead, split, scan kind ----
        // lookahead scans pay when windows commit in full; while they are being truncated (start-up, few overlapping
        // MCs) the scan of a window that then starts elsewhere is wasted
        // (a batch that ran without the tiles' scans and was cut short at a point that needed them: the sparse scans
        // take such points from the next batch on - or, should they be many, the tiles' scans come back)
        sparse_ = c_.allow_nodirty != 0 && few_flagged && flagged > 0 && tiles > 0 &&
                  (trunc_batch == 0 || (nodirty_ && trunc_batch * 4 < wins));
        nodirty_ = c_.allow_nodirty != 0 && tiles > 0 && ((dtiles == 0 && trunc_batch == 0) || sparse_);
        // (switched ON only once the stream is calm - at most one point tile in eight still needs its dirty scan: a window
        // scanned ahead does not see the previous window's commit, and while microclusters are still being promoted that
        // costs a truncated window and two more batches at the start-up window size, ~0.9 ms of a C2 step,
        // profiles/r05_tool_startup_lookahead.txt -; once on it stays on until a window is cut short)
        const bool calm = tiles > 0 && (dtiles * 8 <= tiles || few_flagged);
        // While plain scans run, the batch's first window also runs the pruned chain on 128 of its points (a probe: its
        // results are not used, its sample is): pruned scans come back as soon as the probe says they would pay, instead
        // of being tried on whole batches that cost twice the plain scan when they fail.
        if (!prune_on_ && pr > 0) probe_ok_ = pfu * 12 <= pr;
        const bool probing = c_.prune_applicable != 0 && c_.prune_mode == 1 && !prune_on_ && prev_probe_;
        const bool blind = c_.allow_probe == 0 && o.cursor >= prune_resume_at_;  // (the round-3 rule: try again after a stretch)
        const bool prune_next0 = c_.prune_applicable != 0 && c_.prune_mode != 0 &&
                                ((c_.prune_mode == 2 && prune_resume_at_ != std::numeric_limits<long long>::max()) ||
                                 // on: stays on until a batch completes too many rows (prune_resume_at_ moves ahead of the
                                 // cursor); off: comes back on a probe's word - no blind tries on whole batches
                                 (prune_on_ ? o.cursor >= prune_resume_at_ : ((probing && probe_ok_) || blind)) ||
                                 // a settled stream (no tile needed its dirty scan: nothing created, promoted or moved far)
                                 // is where pruned scans pay: tried at once, whatever the earlier tries said
                                 (nodirty_ && prune_resume_at_ != std::numeric_limits<long long>::max()));
        // (round 6) a table of force_prune_rows rows or more: pruned scans whatever the samples said - behind seeds from the matrix
        // cores and the tight threshold (k_seed16, k_seed_merge with F <= 0) the chain returns exactly what the plain scan returns
        // and costs a fraction of it at that size, in the start-up windows too; below that size the chain's five launches cost
        // what the plain scan costs (75 us per 4 096-point window at C2's 5 000 rows) and the rules above decide as before
        const bool forced = c_.prune_applicable != 0 && c_.prune_mode == 1 && c_.force_prune_rows > 0 && o.m_rows >= c_.force_prune_rows &&
                            prune_resume_at_ != std::numeric_limits<long long>::max();
        const bool prune_next = prune_next0 || forced;
        const bool prune_flip = prune_next != prune_on_;
        prune_on_ = prune_next;
        // (a pending lookahead scan was made for the old split of the table rows / the old kind of scan: restart on a change)
        const bool shard_next = policy_want_shard(c_, o.m_rows, prune_on_);
        const bool shard_flip = shard_next != shard_on_;
        shard_on_ = shard_next;
        // (round 6) ... and while the scan is long enough to be worth a stream of its own.  A PRUNED scan no longer is: with its
        // prefix test on the matrix cores (k_scan_p3) a full window's chain takes 55 us of the window's 180, the two streams share one
        // machine - what runs beside the scan runs that much slower (profiles/r06_tool_window_detail.txt) -, and the scan copies,
        // the carry set and its dirty scan are work the in-place scan does not have: C2 14.2 -> 14.0 ms, the C4 shape 148 -> 154 M
        // points/s with lookahead off.  Split over the ranks of a group the scan's chain ends in an all-gather, which the second
        // stream does hide: lookahead as before.
        const bool short_scan = c_.lookahead != 3 && prune_on_ && !shard_on_ && c_.lookahead_pruned == 0;
        const bool want_la = c_.lookahead == 3 || (c_.lookahead != 2 && trunc_batch == 0 && !unpruned && (la_on_ || calm) && !short_scan);
        probe_gate_ = grew == 0 && !unpruned && want >= std::min(8192, c_.window);
        // Guessed thresholds (k_scan_p with Ctl::tg instead of k_seed + k_seed_merge, which cost as much as the scan they
        // serve): while a mean join distance exists and few points are missed - more than one in sixteen: back to seeds
        // for 2^18 points.  (Split over ranks as well: the list of missed points is derived from the gathered records.)
        const long long missed = o.stat_missed - prev_.stat_missed;
        if (guess_on_ && pts > 0 && missed * 16 > pts) guess_resume_at_ = o.cursor + (1ll << 18);
        const bool guess_was = guess_on_;
        guess_on_ = prune_on_ && c_.allow_guess != 0 && o.tg_ok != 0 && o.cursor >= guess_resume_at_;
        // Lean guessed scans: in a settled stream no point is missed for hundreds of windows, yet every window pays for
        // the machinery that would rescan one - k_missed, the seeded chain's three launches over an empty list, in a
        // group the second all-gather: a fifth of the scan chain's time.  After a batch with guessed thresholds and not one
        // missed point the scans run without it; a point that is missed then keeps its bound,
        // k_decide refuses it (and counts it: stat_missed), the window commits up to it, the batch idles behind it, and the
        // next batch - this rule seeing the count - brings the machinery back.  (allow_guess == 2: never lean.)
        clean_guess_batches_ = (guess_was && guess_on_ && wins > 0 && missed == 0) ? clean_guess_batches_ + 1 : 0;
        lean_on_ = guess_on_ && c_.allow_guess == 1 && clean_guess_batches_ >= 1;
        const bool more = done < c_.n_end;
        const int restart = ((want != win_cfg_ || want_la != la_on_ || o.stall_b > 0 || shard_flip || prune_flip) && more) ? 1 : 0;
        if (restart) {
            win_cfg_ = want;
            la_on_ = want_la;
        }
        // settle quickly at the start of a call and whenever windows are being truncated ... and while the window is
        // held at the start-up size: the end of that phase is only seen at a batch boundary
        batch_windows_ = (trunc_batch > 0 || first_batch_ || want < target || target < win) ? std::max(2, c_.windows_per_sync / 4)
                                                                                            : c_.windows_per_sync;
        if (want < kFastBelow && trunc_batch == 0) batch_windows_ = 2;
        first_batch_ = false;
        // windows that keep stopping short on a small table (input of the sequential-kernel rule)
        const int bad = (trunc_batch > 0 && trunc_batch * 4 >= wins && want <= 1024) ? 1 : 0;
        prev_ = o;
        cc_policy_decision d = decision(restart, bad);
        prev_probe_ = d.probe != 0;
        d.want = want;
        d.wins = wins; d.pts = pts; d.trunc = trunc_batch; d.unk = unk_batch; d.tiles = tiles; d.dtiles = dtiles; d.grew = grew;
        d.prune_rows = (long long)pr; d.prune_full = (long long)pfu;
        return d;
    }

    const cc_policy_carry& carry() const { return k_; }
    bool shard_on() const { return shard_on_; }

private:
    cc_policy_decision decision(int restart, int bad) const
    {
        cc_policy_decision d{};
        d.win_cfg = win_cfg_;
        d.want = win_cfg_;
        d.rounds = rcur_;
        d.batch_windows = batch_windows_;
        d.lookahead = la_on_ ? 1 : 0;
        d.nodirty = nodirty_ ? 1 : 0;
        d.sparse = sparse_ ? 1 : 0;
        // (no probe while the table still grows, most tiles need their dirty scans or the windows are held short: pruned
        // scans cannot pay yet, and on short windows even the probe's three small launches are a few per cent of a batch)
        d.probe = (c_.allow_probe != 0 && c_.prune_applicable != 0 && c_.prune_mode == 1 && !prune_on_ && probe_gate_ &&
                   prune_resume_at_ != std::numeric_limits<long long>::max()) ? 1 : 0;
        d.prune = prune_on_ ? (guess_on_ ? (lean_on_ ? 3 : 2) : 1) : 0;
        d.shard = shard_on_ ? 1 : 0;
        d.restart = restart;
        d.bad = bad;
        d.stalled = stalled_ >= 5 ? 1 : 0;
        return d;
    }

    cc_policy_config c_;
    cc_policy_carry k_;
    cc_policy_obs prev_{};
    int win_cfg_ = 0, rcur_ = 0, batch_windows_ = 2, stalled_ = 0;
    long long prune_resume_at_ = 0, prune_backoff_ = 65536;  // pruned scans are tried again from this point on / stretch after the next failed try
    long long guess_resume_at_ = 0;                          // guessed thresholds are tried again from this point on
    bool guess_on_ = false;
    bool lean_on_ = false;         // guessed thresholds without the list of missed points and their seeded chain
    int clean_guess_batches_ = 0;  // batches in a row with guessed thresholds and no missed point
    bool probe_ok_ = false;    // the last probe of the pruned chain completed few rows: pruned scans would pay
    bool prev_probe_ = false;  // the batch that just ran carried a probe
    bool probe_gate_ = false;  // the last batch's table did not grow and most of its tiles were clean: a probe is worth its cost
    bool la_on_ = false, nodirty_ = false, sparse_ = false, shard_on_ = false, prune_on_ = false, first_batch_ = true;
};

}  // namespace cc
