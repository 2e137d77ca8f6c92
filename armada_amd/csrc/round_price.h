// round_price.h — the indicative gang pricer's node scoring, one node per thread.
//
// Reference: internal/scheduler/scheduling/pricer/node_scheduler.go:41-146 (MinPriceNodeScheduler.Schedule, getJobDetails), preemption_info.go:17-27 (priceOrder).  For ONE
// job and ONE node: static requirements; fits at the evicted priority -> price 0; otherwise EVERY job on the node except the priced gang's own members — any
// priority class, preemptible or not: on a market-driven pool the price decides — ordered by (bid price, age, job id) and preempted one at a time until the job
// fits; the price is the bid of the last victim.  GangPricer.scheduleOnNodes (gang_pricer.go:119-161) evaluates this for every node per gang member: the wide kernel
// (k_price_score: all nodes at once; each thread sorts its node's jobs in its slice of an HBM list laid out like the node -> jobs index, so no node is too full).
#pragma once
#include "round_ctl.h"

struct PriceArgs {
  int32_t job, gang;        // the member being priced; the dense gang id whose members are protected (-1: only the job itself)
  int64_t nowMs;            // start := time.Now() of getJobDetails: ages are now - lease time
  const double* bid;        // [M] job.GetBidPrice(pool) (asched_jobs.bid_price), NULL = 0
};
struct PriceNodeOut { int32_t scheduled, npre; double price; };
struct PriceEntry { int32_t job, pad; double cost; int64_t age; };

DEV bool priceLess(const PriceEntry& a, const PriceEntry& b) {   // priceOrder.Less (preemption_info.go:17-27)
  if (a.cost != b.cost) return a.cost < b.cost;
  if (a.age != b.age) return a.age < b.age;
  return a.job < b.job;
}
// e: this node's slice of the entry list (nodeOff[n+1] - nodeOff[n] entries); preOut (optional): the victims in preemption order
DEV void priceScoreNode(const Dev& d, const PriceArgs& a, const int32_t* nodeOff, const int32_t* nodeJobs, const int64_t* leaseMs, int n, PriceNodeOut* out, int32_t* preOut, PriceEntry* e) {
  const DevCfg& c = d.cfg;
  out->scheduled = 0; out->npre = 0; out->price = 0;
  int job = a.job;
  const uint64_t* mask = d.shapeMask + (size_t)d.jShape[job] * c.W;          // StaticJobRequirementsMet (nodematching.go:161-190) as the shape's static row
  if (!((mask[n >> 6] >> (n & 63)) & 1)) return;
  const int64_t* req = JREQ(d, job);
  int64_t avail[MAXR];
  bool fits = true;
  for (int r = 0; r < MAXR; r++) { avail[r] = r < c.R ? AL(d, c.evLevel, r, n) : 0; if (r < c.R && req[r] > avail[r]) fits = false; }
  if (fits) { out->scheduled = 1; return; }                                   // :55-63
  int m = 0;
  for (int k = nodeOff[n]; k < nodeOff[n + 1]; k++) {                         // node.AllocatedByJobId (:109-146)
    int j = nodeJobs[k];
    if (j == job || (a.gang >= 0 && d.jGang[j] == a.gang)) continue;         // excludedJobIds: the gang's own members
    PriceEntry x;
    x.job = j; x.pad = 0;
    x.cost = a.bid ? a.bid[j] : 0.0;
    x.age = d.jNode0[j] < 0 ? 0 : a.nowMs - leaseMs[j];                      // job.Queued() (placed in this pricing / this round): age 0
    int i = m - 1;                                                            // insertion into the sorted slice
    while (i >= 0 && priceLess(x, e[i])) { e[i + 1] = e[i]; i--; }
    e[i + 1] = x; m++;
  }
  double maxPrice = 0;
  bool ok = false; int used = 0;
  for (int i = 0; i < m; i++) {                                               // :71-80
    const int64_t* jr = JREQ(d, e[i].job);
    bool f = true;
    for (int r = 0; r < c.R; r++) { avail[r] += jr[r]; if (req[r] > avail[r]) f = false; }
    maxPrice = e[i].cost;
    if (preOut) preOut[i] = e[i].job;
    used = i + 1;
    if (f) { ok = true; break; }
  }
  if (!ok) return;
  out->scheduled = 1; out->npre = used; out->price = maxPrice;
}
