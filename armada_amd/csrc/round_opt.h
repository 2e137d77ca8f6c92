// round_opt.h — the experimental fairness optimiser's node scoring, one node per thread.
//
// Reference: internal/scheduler/scheduling/optimiser/node_scheduler.go:42-132 (PreemptingNodeScheduler.Schedule), :134-242
// (getPreemptibleJobDetailsByQueue, populateQueueImpactFields), preemption_info.go:23-91 (the two orderings).  For ONE job and ONE node:
// which of the node's preemptible, non-gang jobs scheduled at a priority not above the job's would have to go — taken in the "ideal" global
// order (per queue: lowest scheduled-at priority, then cheapest, youngest; across queues: priority preemptions first, then the queue whose
// cost after the preemption stays highest) — until the job fits, what that costs (the DRF cost of victims that push their queue to or below
// its fair share) and the largest relative dent in any queue's cost.  FairnessOptimisingGangScheduler.scheduleOnNodes (gang_scheduler.go:80-141)
// evaluates this for EVERY node per job: that loop is the wide kernel here (k_opt_score: all nodes at once, each thread walks its node's job list).
#pragma once
#include "round_ctl.h"

#define OPT_MAXJ 48   // preemptible candidates the per-thread walk of k_opt_score keeps in private memory; a node with more reports overflow (scheduled = -1) and is scored
                      // again with its entries in an HBM scratch list sized by its job count (k_opt_score_big: rare, one thread per such node)

struct OptArgs {
  int32_t job, hasMaxSize;
  int64_t maxSize[MAXR];     // maximumJobSizeToPreempt (0 entries = no limit on that resource, node_scheduler.go:248-268)
  int64_t nowMs;             // start := time.Now() of getPreemptibleJobDetailsByQueue: ages are now - lease time
};
struct OptNodeOut { int32_t scheduled, npre; double cost, impact; };   // nodeSchedulingResult: scheduled, len(jobIdsToPreempt), schedulingCost, maximumQueueImpact
struct OptEntry { int32_t job, queue, sap, ordinal; int64_t age; double cost, wcap; int32_t prioPre, ctpZero; };   // costToPreempt = ctpZero ? 0 : cost

DEV double optRound8(double x) { return round(x * 100000000.0) / 100000000.0; }   // roundFloatHighPrecision (math.Round: half away from zero)

// internalQueueOrder.Less at sort time: every costToPreempt is still 0 (populateQueueImpactFields sorts first, :205), so the order is
// scheduled-at priority, cost, age, job id (preemption_info.go:31-51); entries of different queues are grouped by queue first
DEV bool optInQueueLess(const OptEntry& a, const OptEntry& b) {
  if (a.queue != b.queue) return a.queue < b.queue;
  if (a.sap != b.sap) return a.sap < b.sap;
  if (a.cost != b.cost) return a.cost < b.cost;
  if (a.age != b.age) return a.age < b.age;
  return a.job < b.job;
}
// globalPreemptionOrder.Less (preemption_info.go:62-87)
DEV bool optGlobalLess(const OptEntry& a, const OptEntry& b) {
  if (a.queue == b.queue) return a.ordinal < b.ordinal;
  if (a.prioPre != b.prioPre) return a.prioPre != 0;
  if (a.wcap > b.wcap) return true;
  if (a.wcap == b.wcap) {
    if (a.sap != b.sap) return a.sap < b.sap;
    if (a.cost != b.cost) return a.cost < b.cost;
    if (a.age != b.age) return a.age < b.age;
    return a.job < b.job;
  }
  return false;
}

// PreemptingNodeScheduler.Schedule for (a.job, node n).  qCost[q] = QueueContext.CurrentCost (scheduling_context.go:19-24), fair share =
// demand-capped adjusted fair share, weight: the round's queue state.  preOut (optional): the jobs to preempt, in order.
DEV void optScoreNodeE(const Dev& d, const OptArgs& a, const double* qCost, const int32_t* nodeOff, const int32_t* nodeJobs, const int64_t* leaseMs, int n, OptNodeOut* out, int32_t* preOut,
                       OptEntry* e, int cap) {
  const DevCfg& c = d.cfg;
  out->scheduled = 0; out->npre = 0; out->cost = 0; out->impact = 0;
  int job = a.job;
  const uint64_t* mask = d.shapeMask + (size_t)d.jShape[job] * c.W;          // StaticJobRequirementsMet (nodematching.go:161-190) as the shape's static row
  if (!((mask[n >> 6] >> (n & 63)) & 1)) return;
  const int64_t* req = JREQ(d, job);
  int64_t avail[MAXR];
  bool fits = true;
  for (int r = 0; r < MAXR; r++) { avail[r] = r < c.R ? AL(d, c.evLevel, r, n) : 0; if (r < c.R && req[r] > avail[r]) fits = false; }
  if (fits) { out->scheduled = 1; return; }                                   // :57-64: fits without preemption
  int32_t jobPrio = c.pcPriority[d.jPc[job]];
  int m = 0;
  for (int k = nodeOff[n]; k < nodeOff[n + 1]; k++) {                         // node.AllocatedByJobId (:137-200)
    int j = nodeJobs[k];
    if (!c.pcPreemptible[d.jPc[j]]) continue;
    if (a.hasMaxSize) {                                                       // isTooLargeToEvict (:248-268): any limited resource the job exceeds
      const int64_t* jr = JREQ(d, j);
      bool big = false;
      for (int r = 0; r < c.R; r++) if (a.maxSize[r] != 0 && jr[r] > a.maxSize[r]) big = true;
      if (big) continue;
    }
    if (d.jGang[j] >= 0) continue;
    int32_t sap = d.schedAtPrio[j];
    if (sap == NO_PRIORITY) continue;
    if (sap > jobPrio) continue;
    if (m >= cap) { out->scheduled = -1; return; }                           // more candidates than this call's entry list holds: scored again with a list in HBM
    OptEntry& x = e[m++];
    x.job = j; x.queue = d.jQueue[j]; x.sap = sap; x.ordinal = 0;
    x.age = d.jNode0[j] < 0 ? 0 : a.nowMs - leaseMs[j];                      // job.Queued() (scheduled in this round): age 0
    x.cost = drf(const_cast<Dev&>(d), JREQ(d, j));                            // UnweightedCostFromAllocation(jobResource)
    x.ctpZero = 1; x.wcap = 0; x.prioPre = 0;
  }
  // per queue: order, then the running queue cost (populateQueueImpactFields :203-232)
  for (int i = 1; i < m; i++) { OptEntry x = e[i]; int k = i - 1; while (k >= 0 && optInQueueLess(x, e[k])) { e[k + 1] = e[k]; k--; } e[k + 1] = x; }
  for (int i = 0; i < m;) {
    int q = e[i].queue;
    double updated = qCost[q];
    int ord = 0;
    for (; i < m && e[i].queue == q; i++) {
      updated = optRound8(updated - e[i].cost);
      e[i].wcap = updated / d.qWeight[q];
      if (e[i].sap < jobPrio) { e[i].ctpZero = 1; e[i].prioPre = 1; }
      else if (updated > d.qDc[q]) e[i].ctpZero = 1;
      else e[i].ctpZero = 0;
      e[i].ordinal = ord++;
    }
  }
  for (int i = 1; i < m; i++) { OptEntry x = e[i]; int k = i - 1; while (k >= 0 && optGlobalLess(x, e[k])) { e[k + 1] = e[k]; k--; } e[k + 1] = x; }
  // preempt one job at a time until the job fits (:84-99)
  double total = 0;
  int used = 0; bool ok = false;
  for (int i = 0; i < m; i++) {
    const int64_t* jr = JREQ(d, e[i].job);
    bool f = true;
    for (int r = 0; r < c.R; r++) { avail[r] += jr[r]; if (req[r] > avail[r]) f = false; }
    total += e[i].ctpZero ? 0.0 : e[i].cost;
    if (preOut) preOut[i] = e[i].job;
    used = i + 1;
    if (f) { ok = true; break; }
  }
  if (!ok) return;
  // maximumQueueImpact (:101-113): per queue |sum of the preempted jobs' costs| / CurrentCost
  double impact = 0;
  for (int i = 0; i < used; i++) {
    bool first = true;
    for (int k = 0; k < i; k++) if (e[k].queue == e[i].queue) first = false;
    if (!first) continue;
    double change = 0;
    for (int k = i; k < used; k++) if (e[k].queue == e[i].queue) change -= e[k].cost;   // queueCostChanges[queue] -= cost, in preemption order
    double imp = fabs(change) / qCost[e[i].queue];
    if (imp > impact) impact = imp;
  }
  out->scheduled = 1; out->npre = used; out->cost = total; out->impact = impact;
}
DEV void optScoreNode(const Dev& d, const OptArgs& a, const double* qCost, const int32_t* nodeOff, const int32_t* nodeJobs, const int64_t* leaseMs, int n, OptNodeOut* out, int32_t* preOut) {
  OptEntry e[OPT_MAXJ];
  optScoreNodeE(d, a, qCost, nodeOff, nodeJobs, leaseMs, n, out, preOut, e, OPT_MAXJ);
}
