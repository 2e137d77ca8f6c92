// gpu_round.go — the cgo shim that puts libarmada_sched.so behind the unchanged Go scheduler.
//
// Drop this file into internal/scheduler/scheduling/ of armadaproject/armada (package scheduling) and build with
//
//	CGO_CFLAGS="-I$REPO/include" CGO_LDFLAGS="-L$REPO/armada_amd/csrc -larmada_sched -Wl,-rpath,$REPO/armada_amd/csrc"
//
// FairSchedulingAlgo.SchedulePool (scheduling_algo.go:884-998) then calls GpuRound.Schedule in place of
// NewPreemptingQueueScheduler(...).Schedule(...).  The build image of this repository has no Go toolchain, so this file
// has never been compiled there; the identical C ABI is exercised from Python (armada_amd/binding.py) by every test.
// Every exported C function is declared in include/armada_sched.h next to the Go method it replaces.
//
// Identity: the library works on dense indices.  Node i is nodes[i] of UploadNodes, job j is jobs[j] of UploadJobs, queue q
// is the q-th queue of the sorted queue-name list; the shim keeps the slices to map results back.  Strings never cross the
// boundary: names that the reference orders by (queue names, node ids) are passed as lexicographic ranks, label / taint
// keys and values as interned ids.
package scheduling

/*
#include <stdlib.h>
#include "armada_sched.h"
*/
import "C"

import (
	"context"
	"math"
	"runtime"
	"sort"
	"strconv"
	"sync"
	"time"
	"unsafe"

	"github.com/pkg/errors"
	v1 "k8s.io/api/core/v1"
	k8sResource "k8s.io/apimachinery/pkg/api/resource"

	"github.com/armadaproject/armada/internal/common/armadacontext"
	"github.com/armadaproject/armada/internal/scheduler/configuration"
	"github.com/armadaproject/armada/internal/scheduler/internaltypes"
	"github.com/armadaproject/armada/internal/scheduler/jobdb"
	"github.com/armadaproject/armada/internal/scheduler/nodedb"
	schedulerconstraints "github.com/armadaproject/armada/internal/scheduler/scheduling/constraints"
	schedulercontext "github.com/armadaproject/armada/internal/scheduler/scheduling/context"
	"github.com/armadaproject/armada/internal/scheduler/scheduling/pricer"
)

// interner: strings -> dense ids (label / taint keys and values).
type interner struct {
	ids  map[string]int32
	strs []string // id -> string (ExcludedNodes formats the reference's reason strings from ids)
}

func newInterner() *interner { return &interner{ids: map[string]int32{}} }
func (in *interner) id(s string) int32 {
	if v, ok := in.ids[s]; ok {
		return v
	}
	v := int32(len(in.ids))
	in.ids[s] = v
	in.strs = append(in.strs, s)
	return v
}
func (in *interner) str(id int32) string {
	if id < 0 || int(id) >= len(in.strs) {
		return ""
	}
	return in.strs[id]
}

// rank of every string of xs in lexicographic order (ties cannot occur: names are unique).
func ranks(xs []string) []int32 {
	idx := make([]int, len(xs))
	for i := range idx {
		idx[i] = i
	}
	sort.Slice(idx, func(a, b int) bool { return xs[idx[a]] < xs[idx[b]] })
	out := make([]int32, len(xs))
	for r, i := range idx {
		out[i] = int32(r)
	}
	return out
}

// GpuRound owns one asched handle == one pool: one NodeDb + one SchedulingContext, like nodeDb in SchedulePool.
// One goroutine per GpuRound at a time (the async runner already guarantees a single in-flight run, runner/async.go:34-39);
// handles of different pools are independent and may sit on different GPUs.
type GpuRound struct {
	h        *C.asched_t
	lastCode C.int32_t // return code of the last library call (ErrPeer)
	pool     string
	resNames []string // ResourceListFactory column order (resource_list_factory.go:41-52)
	pcNames  []string // sorted priority-class names == the library's priority-class indices
	pcIndex  map[string]int32
	strs     *interner
	nodes    []*internaltypes.Node
	nodePos  map[string]int32
	jobs     []*jobdb.Job
	classes  map[string]int32 // requirement class key -> index
	// poolConfig.GetDefaultJobTolerations() (scheduling_algo.go:773): part of every requirement class (UploadJobs)
	defaultTolerations []v1.Toleration
	rlf                *internaltypes.ResourceListFactory
	// config.GetMarketConfig(pool) (preempting_queue_scheduler.go:61-62): nil or !Enabled = fair-share scheduling
	market *configuration.MarketSchedulingConfig
}

func effectOf(e v1.TaintEffect) int32 {
	switch e {
	case v1.TaintEffectNoSchedule:
		return C.ASCHED_EFFECT_NO_SCHEDULE
	case v1.TaintEffectPreferNoSchedule:
		return C.ASCHED_EFFECT_PREFER_NO_SCHEDULE
	case v1.TaintEffectNoExecute:
		return C.ASCHED_EFFECT_NO_EXECUTE
	}
	return C.ASCHED_EFFECT_NONE
}

func (g *GpuRound) vec(rl internaltypes.ResourceList) []int64 {
	out := make([]int64, len(g.resNames))
	for i, n := range g.resNames {
		out[i] = rl.GetRawByNameZeroIfMissing(n)
	}
	return out
}

// listFromRaw builds a ResourceList from the library's raw int64 vector with the factory's OWN public constructors (round-3 ADVICE: there is no
// FromInt64Slice upstream and ResourceList.resources is unexported): raw value r of column i is r * 10^scale_i (resource_list_factory.go:41-52, GetScale :151),
// which NewScaledQuantity represents exactly, so the round-up of FromJobResourceListIgnoreUnknown (:99-109) returns r again.
func (g *GpuRound) listFromRaw(raw []int64) internaltypes.ResourceList {
	m := make(map[string]k8sResource.Quantity, len(raw))
	for i, n := range g.resNames {
		if raw[i] == 0 {
			continue
		}
		scale, err := g.rlf.GetScale(n)
		if err != nil {
			continue // a column the factory does not know (cannot happen: resNames is the factory's own column order)
		}
		m[n] = *k8sResource.NewScaledQuantity(raw[i], scale)
	}
	return g.rlf.FromJobResourceListIgnoreUnknown(m)
}

func i32p(x []int32) *C.int32_t {
	if len(x) == 0 {
		return nil
	}
	return (*C.int32_t)(unsafe.Pointer(&x[0]))
}
func i64p(x []int64) *C.int64_t {
	if len(x) == 0 {
		return nil
	}
	return (*C.int64_t)(unsafe.Pointer(&x[0]))
}
func u8p(x []uint8) *C.uint8_t {
	if len(x) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&x[0]))
}
func f64p(x []float64) *C.double {
	if len(x) == 0 {
		return nil
	}
	return (*C.double)(unsafe.Pointer(&x[0]))
}

// NewGpuRound mirrors nodedb.NewNodeDb (nodedb.go:193) + ConfigureScheduling (:386) + the hot-path subset of
// configuration.SchedulingConfig.  resNames is the factory's column order; floating[i] >= 0 marks a floating resource and
// carries the pool's total (floatingresources.GetTotalAvailableForPool), -1 an ordinary one.
func NewGpuRound(cfg configuration.SchedulingConfig, rlf *internaltypes.ResourceListFactory, resNames []string, floating []int64, pool string, device int) (*GpuRound, error) {
	g := &GpuRound{pool: pool, resNames: resNames, strs: newInterner(), pcIndex: map[string]int32{}, classes: map[string]int32{}, rlf: rlf, market: cfg.GetMarketConfig(pool)}
	var pins runtime.Pinner // the config struct points at Go memory for the duration of asched_create
	defer pins.Unpin()
	R := len(resNames)
	col := map[string]int32{}
	for i, n := range resNames {
		col[n] = int32(i)
	}
	var c C.asched_config
	c.num_resources = C.int32_t(R)
	// indexed resources in config order, resolutions in factory units (nodedb.go:274-295)
	idxCol, idxRes := []int32{}, []int64{}
	for _, r := range cfg.IndexedResources {
		idxCol = append(idxCol, col[r.Name])
		scale, err := rlf.GetScale(r.Name)
		if err != nil {
			return nil, err
		}
		idxRes = append(idxRes, r.Resolution.ScaledValue(scale))
	}
	c.num_indexed = C.int32_t(len(idxCol))
	pins.Pin(&idxCol[0]); pins.Pin(&idxRes[0])
	c.indexed_col, c.indexed_resolution = i32p(idxCol), i64p(idxRes)
	// priority classes in name order; away node types as CSR (types.PriorityClass, internal/common/types/scheduling.go:56-76)
	for name := range cfg.PriorityClasses {
		g.pcNames = append(g.pcNames, name)
	}
	sort.Strings(g.pcNames)
	wktIndex := map[string]int32{}
	wktOff, wktKey, wktVal, wktEff := []int32{0}, []int32{}, []int32{}, []int32{}
	for i, t := range cfg.WellKnownNodeTypes {
		wktIndex[t.Name] = int32(i)
		for _, x := range t.Taints {
			val := g.strs.id(x.Value)
			if x.Value == configuration.WildCardWellKnownNodeTypeValue {
				val = -1
			}
			wktKey, wktVal, wktEff = append(wktKey, g.strs.id(x.Key)), append(wktVal, val), append(wktEff, effectOf(x.Effect))
		}
		wktOff = append(wktOff, int32(len(wktKey)))
	}
	pcPrio, pcPre := []int32{}, []uint8{}
	awayOff, awayPrio, awayWkt := []int32{0}, []int32{}, []int32{}
	ntOff, ntWkt, condOff, condRes, condOp := []int32{0}, []int32{}, []int32{0}, []int32{}, []int32{}
	condVal := []int64{}
	for i, name := range g.pcNames {
		pc := cfg.PriorityClasses[name]
		g.pcIndex[name] = int32(i)
		pcPrio = append(pcPrio, pc.Priority)
		pre := uint8(0)
		if pc.Preemptible {
			pre = 1
		}
		pcPre = append(pcPre, pre)
		for _, a := range pc.AwayNodeTypes {
			w := int32(-1)
			if a.WellKnownNodeTypeName != "" {
				w = wktIndex[a.WellKnownNodeTypeName]
			}
			awayPrio, awayWkt = append(awayPrio, a.Priority), append(awayWkt, w)
			for _, nt := range a.NodeTypes { // types.AwayTypeEntry: further node types under resource conditions
				ntWkt = append(ntWkt, wktIndex[nt.Name])
				for _, cd := range nt.Conditions {
					rc, ok := col[cd.Resource]
					if !ok {
						rc = -1
					}
					op := int32(99)
					switch cd.Operator {
					case ">":
						op = C.ASCHED_AWAY_COND_GT
					case "<":
						op = C.ASCHED_AWAY_COND_LT
					case "==":
						op = C.ASCHED_AWAY_COND_EQ
					}
					condRes, condOp, condVal = append(condRes, rc), append(condOp, op), append(condVal, cd.Value.Value())
				}
				condOff = append(condOff, int32(len(condRes)))
			}
			ntOff = append(ntOff, int32(len(ntWkt)))
		}
		awayOff = append(awayOff, int32(len(awayPrio)))
	}
	c.num_priority_classes = C.int32_t(len(g.pcNames))
	pins.Pin(&pcPrio[0]); pins.Pin(&pcPre[0]); pins.Pin(&awayOff[0]); pins.Pin(&wktOff[0]); pins.Pin(&ntOff[0]); pins.Pin(&condOff[0])
	c.pc_priority, c.pc_preemptible = i32p(pcPrio), u8p(pcPre)
	c.pc_away_off, c.away_priority, c.away_well_known = i32p(awayOff), i32p(awayPrio), i32p(awayWkt)
	c.num_well_known_types = C.int32_t(len(cfg.WellKnownNodeTypes))
	c.wkt_taint_off, c.wkt_taint_key, c.wkt_taint_value, c.wkt_taint_effect = i32p(wktOff), i32p(wktKey), i32p(wktVal), i32p(wktEff)
	c.away_nt_off, c.away_nt_well_known, c.away_nt_cond_off = i32p(ntOff), i32p(ntWkt), i32p(condOff)
	c.away_cond_resource, c.away_cond_op, c.away_cond_value = i32p(condRes), i32p(condOp), i64p(condVal)
	for _, s := range [][]int32{awayPrio, awayWkt, wktKey, wktVal, wktEff, ntWkt, condRes, condOp} {
		if len(s) > 0 {
			pins.Pin(&s[0])
		}
	}
	if len(condVal) > 0 {
		pins.Pin(&condVal[0])
	}
	// DRF multipliers (fairness.go:69-89) and the whole-unit size of every column (Quantity.Value() on the job side, nodedb.go:634-635)
	mult, unit := make([]float64, R), make([]int64, R)
	for i, n := range resNames {
		unit[i] = 1
		if scale, err := rlf.GetScale(n); err == nil && scale < 0 {
			unit[i] = int64(math.Pow10(int(-scale)))
		}
	}
	for _, r := range cfg.DominantResourceFairnessResourcesToConsider {
		mult[col[r]] = 1
	}
	for _, r := range cfg.ExperimentalDominantResourceFairnessResourcesToConsider {
		m := r.Multiplier
		if m == 0 {
			m = 1
		}
		mult[col[r.Name]] = m
	}
	pins.Pin(&mult[0]); pins.Pin(&unit[0])
	c.drf_multiplier, c.resource_unit = f64p(mult), i64p(unit)
	// indexed taints / labels (node_type.go:77-109)
	taintKeys, labelKeys := []int32{}, []int32{}
	for _, k := range cfg.IndexedTaints {
		taintKeys = append(taintKeys, g.strs.id(k))
	}
	for _, k := range cfg.IndexedNodeLabels {
		labelKeys = append(labelKeys, g.strs.id(k))
	}
	c.num_indexed_taints, c.num_indexed_labels = C.int32_t(len(taintKeys)), C.int32_t(len(labelKeys))
	if len(taintKeys) > 0 {
		pins.Pin(&taintKeys[0])
	}
	if len(labelKeys) > 0 {
		pins.Pin(&labelKeys[0])
	}
	c.indexed_taint_keys, c.indexed_label_keys = i32p(taintKeys), i32p(labelKeys)
	b := func(x bool) C.uint8_t {
		if x {
			return 1
		}
		return 0
	}
	c.prefer_large_job_ordering = b(cfg.EnablePreferLargeJobOrdering)
	c.preempt_cross_pool_jobs_first = b(cfg.GetPreemptCrossPoolJobsFirst(pool)) // preempting_queue_scheduler.go:61-80
	c.protected_fraction_of_fair_share = C.double(cfg.GetProtectedFractionOfFairShare(pool))
	c.protect_uncapped_adjusted_fair_share = b(cfg.GetProtectUncappedAdjustedFairShare(pool))
	c.max_queue_lookback = C.uint32_t(cfg.MaxQueueLookback)
	// MaximumResourceFractionToSchedule (per pool override first, constraints.go:199-216); +Inf = uncapped
	frac := make([]float64, R)
	for i := range frac {
		frac[i] = math.Inf(1)
	}
	limits := cfg.MaximumResourceFractionToSchedule
	if byPool, ok := cfg.MaximumResourceFractionToScheduleByPool[pool]; ok {
		limits = byPool
	}
	for n, f := range limits {
		if i, ok := col[n]; ok {
			frac[i] = f
		}
	}
	pins.Pin(&frac[0])
	c.max_fraction_to_schedule = f64p(frac)
	if len(floating) == R {
		pins.Pin(&floating[0])
		c.floating_resource_limit = i64p(floating)
		c.floating_counts_in_total = 1 // scheduling_algo.go:890
	}
	c.max_new_job_scheduling_duration_ns = C.int64_t(cfg.MaxNewJobSchedulingDuration.Nanoseconds())
	c.max_new_job_scheduling_duration_per_queue_ns = C.int64_t(cfg.MaxNewJobSchedulingDurationPerQueue.Nanoseconds())
	c.device = C.int32_t(device)
	g.h = C.asched_create(&c)
	if g.h == nil {
		return nil, errors.New("asched_create failed: no gfx950 device, or a configuration the library does not represent")
	}
	return g, nil
}

func (g *GpuRound) Close() {
	if g.h != nil {
		C.asched_destroy(g.h)
		g.h = nil
	}
}

func (g *GpuRound) check(rc C.int32_t) error {
	g.lastCode = rc
	if rc == 0 {
		return nil
	}
	return errors.Errorf("asched error %d: %s", int(rc), C.GoString(C.asched_last_error(g.h)))
}

// UploadNodes == nodeDb.CreateAndInsertWithJobDbJobsWithTxn for every node of the pool (nodedb.go:57-75): SoA, row-major [n][R].
func (g *GpuRound) UploadNodes(nodes []*internaltypes.Node) error {
	n, R := len(nodes), len(g.resNames)
	g.nodes, g.nodePos = nodes, make(map[string]int32, n)
	ids := make([]string, n)
	index := make([]uint64, n)
	total, alloc := make([]int64, n*R), make([]int64, n*R)
	unsched, over := make([]uint8, n), make([]uint8, n)
	tOff, tKey, tVal, tEff := make([]int32, 1, n+1), []int32{}, []int32{}, []int32{}
	lOff, lKey, lVal := make([]int32, 1, n+1), []int32{}, []int32{}
	{ // taint keys first, in lexicographic order: the library reports the FIRST untolerated taint of a node type in ascending key id, the reference in ascending key string (node_type.go:87-97)
		keys := map[string]bool{}
		for _, node := range nodes {
			for _, t := range node.GetTaints() {
				keys[t.Key] = true
			}
		}
		sorted := make([]string, 0, len(keys))
		for k := range keys {
			sorted = append(sorted, k)
		}
		sort.Strings(sorted)
		for _, k := range sorted {
			g.strs.id(k)
		}
	}
	for i, node := range nodes {
		ids[i], index[i] = node.GetId(), node.GetIndex()
		g.nodePos[node.GetId()] = int32(i)
		copy(total[i*R:], g.vec(node.GetTotalResources()))
		copy(alloc[i*R:], g.vec(node.GetAllocatableResources()))
		if node.IsUnschedulable() {
			unsched[i] = 1
		}
		if node.IsOverAllocated() {
			over[i] = 1
		}
		for _, t := range node.GetTaints() {
			if t.Key == "node.kubernetes.io/unschedulable" {
				continue // the library adds it from `unschedulable` (node.go:127-129)
			}
			tKey, tVal, tEff = append(tKey, g.strs.id(t.Key)), append(tVal, g.strs.id(t.Value)), append(tEff, effectOf(t.Effect))
		}
		tOff = append(tOff, int32(len(tKey)))
		labels := node.GetLabels()
		keys := make([]string, 0, len(labels))
		for k := range labels {
			keys = append(keys, k)
		}
		sort.Strings(keys)
		for _, k := range keys {
			lKey, lVal = append(lKey, g.strs.id(k)), append(lVal, g.strs.id(labels[k]))
		}
		lOff = append(lOff, int32(len(lKey)))
	}
	rank := ranks(ids)
	var pins runtime.Pinner
	defer pins.Unpin()
	var in C.asched_nodes
	in.n = C.int32_t(n)
	if n > 0 {
		for _, p := range []interface{}{&index[0], &rank[0], &total[0], &alloc[0], &unsched[0], &over[0], &tOff[0], &lOff[0]} {
			pins.Pin(p)
		}
		in.index, in.id_rank = (*C.uint64_t)(unsafe.Pointer(&index[0])), i32p(rank)
		in.total, in.allocatable = i64p(total), i64p(alloc)
		in.unschedulable, in.over_allocated = u8p(unsched), u8p(over)
		in.taint_off, in.label_off = i32p(tOff), i32p(lOff)
	}
	for _, s := range [][]int32{tKey, tVal, tEff, lKey, lVal} {
		if len(s) > 0 {
			pins.Pin(&s[0])
		}
	}
	in.taint_key, in.taint_value, in.taint_effect = i32p(tKey), i32p(tVal), i32p(tEff)
	in.label_key, in.label_value = i32p(lKey), i32p(lVal)
	if err := g.uploadLabelValueInts(); err != nil {
		return err
	}
	return g.check(C.asched_nodes_upsert(g.h, &in))
}

// classOf interns the static requirements of a job — tolerations, node selector, required node affinity — into a requirement class
// (the part of the scheduling key, internaltypes/podutils.go:52-72, that is not requests or priority class).
type reqClass struct {
	tol  [][4]int32
	sel  [][2]int32
	aff  [][][3]interface{} // terms -> expressions -> (key, op, values)
	hasA bool
}

// UploadJobs registers the jobDb view of the round: every non-terminal job of the pool, queued or running (the accessors
// calculateJobSchedulingInfo reads, scheduling_algo.go:591-698).  queueIndex maps queue names to dense indices (name order).
func (g *GpuRound) UploadJobs(jobs []*jobdb.Job, queueIndex map[string]int32) error {
	m, R := len(jobs), len(g.resNames)
	// The ABI's contract (include/armada_sched.h, asched_jobs): a job's row IS its rank in job-id order — the final tie-break of SchedulingOrderCompare
	// (jobdb/comparison.go:99-105), of MarketSchedulingOrderCompare (:160-168) and of the pricer's priceOrder (pricer/node_scheduler.go).  Callers hand jobs over
	// in whatever order they hold them, so the shim sorts (a copy) by id before anything is indexed by row (round-3 ADVICE).
	jobs = append([]*jobdb.Job(nil), jobs...)
	sort.SliceStable(jobs, func(a, b int) bool { return jobs[a].Id() < jobs[b].Id() })
	g.jobs = jobs
	queue, pc, reqClassIdx := make([]int32, m), make([]int32, m), make([]int32, m)
	qprio := make([]uint32, m)
	submit, runTs := make([]int64, m), make([]int64, m)
	req := make([]int64, m*R)
	gangId, gangCard, gangUni := make([]int32, m), make([]int32, m), make([]int32, m)
	node, runPrio := make([]int32, m), make([]int32, m)
	away, anyAway := make([]uint8, m), false
	gangIds := map[string]int32{}
	var classes []reqClass
	classKey := func(j *jobdb.Job) (string, reqClass) {
		var rc reqClass
		key := ""
		// the pool's default job tolerations (SchedulingOptions.DefaultTolerations, nodedb.go:383-393) are appended to every job's tolerations for
		// every node selection (:561-562): folded into the class here, once (g.defaultTolerations = poolConfig.GetDefaultJobTolerations())
		for _, t := range append(append([]v1.Toleration{}, j.Tolerations()...), g.defaultTolerations...) {
			k := int32(-1)
			if t.Key != "" {
				k = g.strs.id(t.Key)
			}
			op := int32(C.ASCHED_TOLERATION_OP_EQUAL)
			if t.Operator == v1.TolerationOpExists {
				op = C.ASCHED_TOLERATION_OP_EXISTS
			}
			rc.tol = append(rc.tol, [4]int32{k, op, g.strs.id(t.Value), effectOf(t.Effect)})
			key += "t" + t.Key + "\x00" + string(t.Operator) + "\x00" + t.Value + "\x00" + string(t.Effect) + "\x01"
		}
		sel := j.NodeSelector()
		keys := make([]string, 0, len(sel))
		for k := range sel {
			keys = append(keys, k)
		}
		sort.Strings(keys)
		for _, k := range keys {
			rc.sel = append(rc.sel, [2]int32{g.strs.id(k), g.strs.id(sel[k])})
			key += "s" + k + "\x00" + sel[k] + "\x01"
		}
		if a := j.Affinity(); a != nil && a.NodeAffinity != nil && a.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution != nil {
			rc.hasA = true
			for _, term := range a.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution.NodeSelectorTerms {
				var exprs [][3]interface{}
				key += "a("
				for _, e := range term.MatchExpressions {
					vals := make([]int32, len(e.Values))
					for i, v := range e.Values {
						vals[i] = g.strs.id(v)
					}
					exprs = append(exprs, [3]interface{}{g.strs.id(e.Key), string(e.Operator), vals})
					key += e.Key + "\x00" + string(e.Operator) + "\x00"
					for _, v := range e.Values {
						key += v + "\x02"
					}
					key += "\x01"
				}
				key += ")"
				rc.aff = append(rc.aff, exprs)
			}
		}
		return key, rc
	}
	if _, ok := g.classes[""]; !ok { // class 0 must exist: no tolerations, no selector
		g.classes[""] = 0
	}
	classes = make([]reqClass, len(g.classes))
	for i, j := range jobs {
		// a job whose queue has no context in this pool keeps queue -1: the library treats it like sctx.QueueContextExists(job) == false
		// ("invalid_queue": never evicted, not part of any aggregate) instead of charging it to queue 0
		if qi, ok := queueIndex[j.Queue()]; ok {
			queue[i] = qi
		} else {
			queue[i] = -1
		}
		pc[i] = g.pcIndex[j.PriorityClassName()]
		qprio[i] = j.Priority()
		submit[i] = j.SubmitTime().UnixNano()
		copy(req[i*R:], g.vec(j.AllResourceRequirements()))
		key, rc := classKey(j)
		ci, ok := g.classes[key]
		if !ok {
			ci = int32(len(g.classes))
			g.classes[key] = ci
		}
		for int(ci) >= len(classes) {
			classes = append(classes, reqClass{})
		}
		classes[ci] = rc
		reqClassIdx[i] = ci
		gangId[i], gangCard[i], gangUni[i] = -1, 1, -1
		if gi := j.GetGangInfo(); gi.IsGang() {
			k := j.Queue() + "\x00" + gi.Id()
			id, ok := gangIds[k]
			if !ok {
				id = int32(len(gangIds))
				gangIds[k] = id
			}
			gangId[i], gangCard[i] = id, int32(gi.Cardinality())
			if gi.NodeUniformity() != "" {
				gangUni[i] = g.strs.id(gi.NodeUniformity())
			}
		}
		node[i] = -1
		// A run in another pool on one of this NodeDb's nodes is a cross-pool away job (context.IsHomeJob false): it is accounted against the
		// "<queue>-away" context (queueIndex holds those beside the home contexts: scheduling_algo.go:840-867) and flagged for the library, which
		// binds it at CrossPoolPriority and orders it after home jobs (include/armada_sched.h asched_jobs.away).
		if run := j.LatestRun(); !j.Queued() && run != nil {
			if p, ok := g.nodePos[run.NodeId()]; ok {
				if run.Pool() != g.pool {
					away[i] = 1
					anyAway = true
					// (the node evictor never takes such a job: pqs.go:102-104; the library knows from the flag)
					if qi, ok := queueIndex[schedulercontext.CalculateAwayQueueName(j.Queue())]; ok {
						queue[i] = qi
					} else {
						queue[i] = -1
					}
				}
				node[i] = p
				runTs[i] = j.ActiveRunTimestamp()
				if sp := run.ScheduledAtPriority(); sp != nil {
					runPrio[i] = *sp
				} else {
					runPrio[i] = j.PriorityClass().Priority
				}
			}
		}
	}
	// requirement classes as CSR
	nc := len(classes)
	tolOff, selOff, affTermOff := make([]int32, 1, nc+1), make([]int32, 1, nc+1), make([]int32, 1, nc+1)
	var tolKey, tolOp, tolVal, tolEff, selKey, selVal, exprKey, exprOp, values []int32
	exprOff, valueOff := []int32{0}, []int32{0}
	hasAff := make([]uint8, nc)
	anyAff := false
	ops := map[string]int32{"In": C.ASCHED_AFFINITY_OP_IN, "NotIn": C.ASCHED_AFFINITY_OP_NOT_IN, "Exists": C.ASCHED_AFFINITY_OP_EXISTS, "DoesNotExist": C.ASCHED_AFFINITY_OP_DOES_NOT_EXIST,
		"Gt": C.ASCHED_AFFINITY_OP_GT, "Lt": C.ASCHED_AFFINITY_OP_LT}
	for ci, rc := range classes {
		for _, t := range rc.tol {
			tolKey, tolOp, tolVal, tolEff = append(tolKey, t[0]), append(tolOp, t[1]), append(tolVal, t[2]), append(tolEff, t[3])
		}
		tolOff = append(tolOff, int32(len(tolKey)))
		for _, s := range rc.sel {
			selKey, selVal = append(selKey, s[0]), append(selVal, s[1])
		}
		selOff = append(selOff, int32(len(selKey)))
		if rc.hasA {
			hasAff[ci], anyAff = 1, true
			for _, term := range rc.aff {
				for _, e := range term {
					op, ok := ops[e[1].(string)]
					if !ok {
						op = 99 // an operator v1.NodeSelectorOperator does not define: the library answers ASCHED_ERR_UNSUPPORTED and the caller keeps this round on the Go NodeDb
					}
					exprKey, exprOp = append(exprKey, e[0].(int32)), append(exprOp, op)
					values = append(values, e[2].([]int32)...)
					valueOff = append(valueOff, int32(len(values)))
				}
				exprOff = append(exprOff, int32(len(exprKey)))
			}
		}
		affTermOff = append(affTermOff, int32(len(exprOff)-1))
	}
	var pins runtime.Pinner
	defer pins.Unpin()
	pin32 := func(s []int32) *C.int32_t {
		if len(s) > 0 {
			pins.Pin(&s[0])
		}
		return i32p(s)
	}
	var cls C.asched_req_classes
	cls.n = C.int32_t(nc)
	cls.tol_off, cls.tol_key, cls.tol_op, cls.tol_value, cls.tol_effect = pin32(tolOff), pin32(tolKey), pin32(tolOp), pin32(tolVal), pin32(tolEff)
	cls.sel_off, cls.sel_key, cls.sel_value = pin32(selOff), pin32(selKey), pin32(selVal)
	if anyAff {
		pins.Pin(&hasAff[0])
		cls.has_affinity = u8p(hasAff)
		cls.aff_term_off, cls.aff_expr_off, cls.aff_expr_key, cls.aff_expr_op = pin32(affTermOff), pin32(exprOff), pin32(exprKey), pin32(exprOp)
		cls.aff_value_off, cls.aff_values = pin32(valueOff), pin32(values)
	}
	var in C.asched_jobs
	in.m = C.int32_t(m)
	if m > 0 {
		pins.Pin(&qprio[0]); pins.Pin(&submit[0]); pins.Pin(&runTs[0]); pins.Pin(&req[0])
		in.queue, in.pc, in.req_class = pin32(queue), pin32(pc), pin32(reqClassIdx)
		in.queue_priority = (*C.uint32_t)(unsafe.Pointer(&qprio[0]))
		in.submit_time, in.run_timestamp, in.req = i64p(submit), i64p(runTs), i64p(req)
		in.gang_id, in.gang_cardinality, in.gang_uniformity_label = pin32(gangId), pin32(gangCard), pin32(gangUni)
		in.node, in.scheduled_at_priority = pin32(node), pin32(runPrio)
		if anyAway {
			pins.Pin(&away[0])
			in.away = (*C.uint8_t)(unsafe.Pointer(&away[0]))
		}
		if g.market != nil && g.market.Enabled { // job.GetBidPrice(pool) as the job stands now (jobdb/job.go:459-481): queued / running / non-preemptible-running bid
			bid := make([]float64, m)
			for i, job := range jobs {
				bid[i] = job.GetBidPrice(g.pool)
			}
			pins.Pin(&bid[0])
			in.bid_price = f64p(bid)
		}
	}
	if err := g.uploadLabelValueInts(); err != nil { // Gt / Lt compare integers: every interned string that parses, before the masks are built
		return err
	}
	return g.check(C.asched_jobs_set(g.h, &in, &cls))
}

// uploadLabelValueInts tells the library which interned strings are integers (strconv.ParseInt(v, 10, 64), what
// labels.Requirement.Matches does for Gt / Lt) — asched_set_label_value_ints.  Called before asched_jobs_set / asched_nodes_upsert
// evaluate affinities (the static masks are rebuilt by whichever of the two comes last).
func (g *GpuRound) uploadLabelValueInts() error {
	var ids []int32
	var ints []int64
	for s, id := range g.strs.ids {
		if v, err := strconv.ParseInt(s, 10, 64); err == nil {
			ids, ints = append(ids, id), append(ints, v)
		}
	}
	var pins runtime.Pinner
	defer pins.Unpin()
	if len(ids) > 0 {
		pins.Pin(&ids[0])
		pins.Pin(&ints[0])
	}
	return g.check(C.asched_set_label_value_ints(g.h, C.int32_t(len(ids)), i32p(ids), i64p(ints)))
}

// Schedule == PreemptingQueueScheduler.Schedule (preempting_queue_scheduler.go:86-289) for one pool.  sctx supplies weights, limiters
// (evaluated at sctx.Started, constraints.go:136-157), penalties and cordon flags; queuedJobs[queue] the queue's queued job indices in
// jobdb order (QueuedJobsIterator, jobiteration.go:138-176).  Allocation and demand per queue are derived by the library from the job
// table (asched_round_prepare with NULL aggregates == calculateJobSchedulingInfo + constructSchedulingContext).
// The context's deadline becomes the round's hard timeout and ctx.Done() cancels a round in flight (queue_scheduler.go:105-112).
func (g *GpuRound) Schedule(ctx *armadacontext.Context, sctx *schedulercontext.SchedulingContext, cordoned map[string]bool,
	perQueuePcLimitFraction map[string]map[string]map[string]float64, queuedJobs map[string][]int32,
) (*SchedulingResult, error) {
	names := make([]string, 0, len(sctx.QueueSchedulingContexts))
	for name := range sctx.QueueSchedulingContexts {
		names = append(names, name)
	}
	sort.Strings(names)
	Q, R, npc := len(names), len(g.resNames), len(g.pcNames)
	nameRank := make([]int32, Q)
	weight, qTokens := make([]float64, Q), make([]float64, Q)
	qBurst := make([]int64, Q)
	qInf, cord := make([]uint8, Q), make([]uint8, Q)
	penalty := make([]int64, Q*R)
	limits := make([]float64, Q*npc*R)
	for i := range limits {
		limits[i] = math.Inf(1)
	}
	off := make([]int32, Q+1)
	var flat []int32
	for q, name := range names {
		qctx := sctx.QueueSchedulingContexts[name]
		nameRank[q] = int32(q) // names are sorted
		weight[q] = qctx.Weight
		qTokens[q] = qctx.Limiter.TokensAt(sctx.Started)
		qBurst[q] = int64(qctx.Limiter.Burst())
		if math.IsInf(float64(qctx.Limiter.Limit()), 1) {
			qInf[q] = 1
		}
		if cordoned[name] {
			cord[q] = 1
		}
		copy(penalty[q*R:], g.vec(qctx.ShortJobPenalty))
		for pcName, byRes := range perQueuePcLimitFraction[name] {
			for res, f := range byRes {
				for r, n := range g.resNames {
					if n == res {
						limits[(q*npc+int(g.pcIndex[pcName]))*R+r] = f
					}
				}
			}
		}
		flat = append(flat, queuedJobs[name]...)
		off[q+1] = int32(len(flat))
	}
	var pins runtime.Pinner
	defer pins.Unpin()
	var in C.asched_queues
	in.q = C.int32_t(Q)
	if Q > 0 {
		for _, p := range []interface{}{&nameRank[0], &weight[0], &qTokens[0], &qBurst[0], &qInf[0], &cord[0], &penalty[0], &limits[0], &off[0]} {
			pins.Pin(p)
		}
		in.name_rank, in.weight = i32p(nameRank), f64p(weight)
		in.short_job_penalty, in.cordoned, in.pc_resource_limit_fraction = i64p(penalty), u8p(cord), f64p(limits)
		in.queue_tokens, in.queue_burst, in.queue_rate_inf = f64p(qTokens), i64p(qBurst), u8p(qInf)
		in.queued_off = i32p(off)
	}
	if len(flat) > 0 {
		pins.Pin(&flat[0])
	}
	in.queued_jobs = i32p(flat)
	in.global_tokens = C.double(sctx.Limiter.TokensAt(sctx.Started))
	in.global_burst = C.int64_t(sctx.Limiter.Burst())
	if math.IsInf(float64(sctx.Limiter.Limit()), 1) {
		in.global_rate_inf = 1
	}
	if sctx.FairsharePreemptionLimiter != nil { // context/scheduling.go:508-528
		in.has_fairshare_preemption_limiter = 1
		in.fairshare_preemption_tokens = C.double(sctx.FairsharePreemptionLimiter.TokensAt(sctx.Started))
	}
	{ // market-driven pool (pqs.go:61-62): evict-everything evictor, price-ordered iterators, spot price (asched_set_market); queuedJobs then come in jobdb.PriceOrder
		var mc C.asched_market_config
		if g.market != nil && g.market.Enabled {
			mc.enabled = 1
			mc.spot_price_cutoff = C.double(g.market.SpotPriceCutoff)
		}
		if err := g.check(C.asched_set_market(g.h, &mc)); err != nil {
			return nil, err
		}
	}
	if err := g.check(C.asched_round_prepare(g.h, &in)); err != nil {
		return nil, err
	}
	// hard timeout + cancellation: the library polls a host-mapped word from inside the round kernel
	// (asched_set_deadline counts from the start of asched_schedule_round: what is left of the context's deadline NOW, not since sctx.Started)
	secs := 0.0
	if dl, ok := ctx.Deadline(); ok {
		secs = math.Max(time.Until(dl).Seconds(), 1e-6)
	}
	if err := g.check(C.asched_set_deadline(g.h, C.double(secs))); err != nil {
		return nil, err
	}
	done := make(chan struct{})
	cancelled := false
	var wg sync.WaitGroup
	wg.Add(1)
	go func() { // asched_cancel may be called from any thread
		defer wg.Done()
		select {
		case <-ctx.Done():
			select {
			case <-done: // the round is over already: a late cancel would hit the NEXT round on this handle
			default:
				cancelled = true
				C.asched_cancel(g.h)
			}
		case <-done:
		}
	}()
	var out C.asched_round_result
	rc := C.asched_schedule_round(g.h, &out)
	close(done)
	wg.Wait() // the goroutine never touches g.h after Schedule returns (Close may follow)
	if cancelled {
		// Whatever rc is: the round has returned, so a request that is still pending belongs to THIS context.  (The library's own deadline is the same instant as
		// the context's; it can end the round and clear the word a moment before the watcher goroutine stores its request — round-3 ADVICE — which would make the
		// next round on this handle fail at once.)
		C.asched_cancel_clear(g.h)
	}
	if rc == C.ASCHED_ERR_TIMEOUT {
		err := ctx.Err()
		if err == nil { // the library's own deadline fired a moment before the context's
			err = context.DeadlineExceeded
		}
		sctx.TerminationReason = "hard timeout: " + err.Error()
		return nil, err // the caller maps this to PoolSchedulingTerminationReasonTimeout (scheduling_algo.go:262-270)
	}
	if err := g.check(rc); err != nil {
		return nil, err // error => round discarded, nothing applied (scheduling_algo.go:262-285)
	}
	if g.market != nil && g.market.Enabled { // sctx.SpotPrice, qctx.BillableResource, qctx.BillablePriceOverride (queue_scheduler.go:177-203)
		var mr C.asched_market_outcome
		if err := g.check(C.asched_market_result(g.h, &mr)); err != nil {
			return nil, err
		}
		if mr.has_spot_price != 0 {
			p := float64(mr.spot_price)
			sctx.SpotPrice = &p
		}
		bill := unsafe.Slice((*int64)(unsafe.Pointer(mr.queue_billable_resource)), Q*R)
		over := unsafe.Slice((*float64)(unsafe.Pointer(mr.queue_billable_price_override)), Q)
		has := unsafe.Slice((*uint8)(unsafe.Pointer(mr.queue_has_price_override)), Q)
		for q, name := range names {
			qctx := sctx.QueueSchedulingContexts[name]
			if mr.has_spot_price != 0 {
				qctx.BillableResource = g.listFromRaw(bill[q*R : (q+1)*R])
			}
			if has[q] != 0 {
				v := over[q]
				qctx.BillablePriceOverride = &v
			}
		}
	}
	return g.buildResult(sctx, &out), nil
}

// PriceGang == pricer.GangPricer.Price (pricer/gang_pricer.go:48-117) on the NodeDb as the round left it; jobs are rows of the uploaded job set (the synthetic
// jobs MarketDrivenIndicativePricer builds from a configuration.GangDefinition are uploaded with the others, queued nowhere).  The limit and deadline checks of
// market_driven_indicative_pricer.go:64-118 stay with the caller.
func (g *GpuRound) PriceGang(jobs []int32, now time.Time) (pricer.GangPricingResult, error) {
	var out C.asched_gang_price
	if err := g.check(C.asched_price_gang(g.h, C.int32_t(len(jobs)), i32p(jobs), C.int64_t(now.UnixMilli()), &out)); err != nil {
		return pricer.GangPricingResult{}, err
	}
	reason := ""
	switch int(out.reason) {
	case C.ASCHED_REASON_JOB_DOES_NOT_FIT:
		reason = schedulerconstraints.JobDoesNotFitUnschedulableReason
	case C.ASCHED_REASON_GANG_DOES_NOT_FIT:
		reason = schedulerconstraints.GangDoesNotFitUnschedulableReason
	case C.ASCHED_PRICE_REASON_LABEL_NOT_INDEXED:
		reason = pricer.GangUniformityLabelIsNotIndexedUnschedulableReason
	case C.ASCHED_PRICE_REASON_NO_NODES_WITH_LABEL:
		reason = pricer.GangNoNodesWithUniformityLabelUnschedulableReason
	}
	return pricer.GangPricingResult{Evaluated: out.evaluated != 0, Schedulable: out.schedulable != 0, Price: float64(out.price), UnschedulableReason: reason}, nil
}

// buildResult turns the flat result into the jctx lists SchedulingResult carries (result.go:96-107).  Result buffers are owned by the
// handle until the next round_prepare: everything is copied here.
func (g *GpuRound) buildResult(sctx *schedulercontext.SchedulingContext, out *C.asched_round_result) *SchedulingResult {
	ns, np := int(out.num_scheduled), int(out.num_preempted)
	sj := unsafe.Slice((*int32)(unsafe.Pointer(out.scheduled_job)), ns)
	sn := unsafe.Slice((*int32)(unsafe.Pointer(out.scheduled_node)), ns)
	sp := unsafe.Slice((*int32)(unsafe.Pointer(out.scheduled_priority)), ns)
	pj := unsafe.Slice((*int32)(unsafe.Pointer(out.preempted_job)), np)
	pn := unsafe.Slice((*int32)(unsafe.Pointer(out.preempted_node)), np)
	res := &SchedulingResult{SchedulingContext: sctx}
	for i := 0; i < ns; i++ {
		jctx := schedulercontext.JobSchedulingContextFromJob(g.jobs[sj[i]])
		node := g.nodes[sn[i]]
		jctx.PodSchedulingContext = &schedulercontext.PodSchedulingContext{
			Created:             sctx.Started,
			NodeId:              node.GetId(),
			ScheduledAtPriority: sp[i], // nodeDb.GetScheduledAtPriority (nodedb.go:315)
			NumNodes:            len(g.nodes),
		}
		res.ScheduledJobs = append(res.ScheduledJobs, jctx)
	}
	for i := 0; i < np; i++ {
		jctx := schedulercontext.JobSchedulingContextFromJob(g.jobs[pj[i]])
		jctx.AssignedNode = g.nodes[pn[i]] // the node the job is preempted from (preempting_queue_scheduler.go:255-266)
		res.PreemptedJobs = append(res.PreemptedJobs, jctx)
	}
	// per-queue allocation by priority class, fair shares and the limiters' token counts after the round feed metrics / reports
	Q, R, npc := len(sctx.QueueSchedulingContexts), len(g.resNames), len(g.pcNames)
	alloc := unsafe.Slice((*int64)(unsafe.Pointer(out.queue_allocated_by_pc)), Q*npc*R)
	fair := unsafe.Slice((*float64)(unsafe.Pointer(out.queue_fair_share)), Q)
	dcafs := unsafe.Slice((*float64)(unsafe.Pointer(out.queue_demand_capped_adjusted_fair_share)), Q)
	ucafs := unsafe.Slice((*float64)(unsafe.Pointer(out.queue_uncapped_adjusted_fair_share)), Q)
	names := make([]string, 0, Q)
	for name := range sctx.QueueSchedulingContexts {
		names = append(names, name)
	}
	sort.Strings(names)
	for q, name := range names {
		qctx := sctx.QueueSchedulingContexts[name]
		qctx.FairShare, qctx.DemandCappedAdjustedFairShare, qctx.UncappedAdjustedFairShare = fair[q], dcafs[q], ucafs[q]
		_ = alloc[q*npc*R] // AllocatedByPriorityClass: rebuilt through the ResourceListFactory by the caller if it reports it
	}
	sctx.TerminationReason = terminationReasonString(int(out.termination_reason))
	return res
}

// the strings of constraints.go:25-58 for the ASCHED_REASON_* codes
func terminationReasonString(r int) string {
	switch r {
	case C.ASCHED_REASON_MAX_RESOURCES_SCHEDULED:
		return "maximum resources scheduled"
	case C.ASCHED_REASON_GLOBAL_RATE_LIMIT:
		return "global scheduling rate limit exceeded"
	case C.ASCHED_REASON_GLOBAL_NEW_JOB_DURATION:
		return "global new job scheduling duration exceeded"
	case C.ASCHED_REASON_NO_REMAINING_CANDIDATES:
		return "no remaining candidate jobs"
	}
	return ""
}

// ---- NodeDb-level calls, 1:1 with nodedb.go, for callers that drive the NodeDb directly (GangScheduler, Evictor, submit check)

func (g *GpuRound) Txn() error    { return g.check(C.asched_txn_begin(g.h)) }  // nodeDb.Txn(true) :353
func (g *GpuRound) Commit() error { return g.check(C.asched_txn_commit(g.h)) }
func (g *GpuRound) Abort() error  { return g.check(C.asched_txn_abort(g.h)) }

// ScheduleManyWithTxn (nodedb.go:417-462): ok, the members' nodes (-1 = none) and the jobs preempted to make room.
func (g *GpuRound) ScheduleMany(jobs []int32) (bool, []int32, []int32, error) {
	n := len(jobs)
	out := make([]C.asched_pod_result, n)
	pre := make([]int32, 65536)
	var ok, npre C.int32_t
	rc := C.asched_schedule_many(g.h, C.int32_t(n), i32p(jobs), nil, &out[0], &ok, i32p(pre), C.int32_t(len(pre)), &npre)
	if err := g.check(rc); err != nil {
		return false, nil, nil, err
	}
	nodes := make([]int32, n)
	for i := range out {
		nodes[i] = int32(out[i].node)
	}
	return ok != 0, nodes, pre[:int(npre)], nil
}

func (g *GpuRound) Bind(job, node, priority int32) error { // BindJobToNode + UpsertWithTxn (:1046-1068, :1164)
	return g.check(C.asched_bind(g.h, C.int32_t(job), C.int32_t(node), C.int32_t(priority)))
}
func (g *GpuRound) Evict(job, node int32) error { // EvictJobsFromNode (:1079)
	return g.check(C.asched_evict(g.h, C.int32_t(job), C.int32_t(node)))
}
func (g *GpuRound) Unbind(job, node int32) error { // UnbindJobFromNode (:1108)
	return g.check(C.asched_unbind(g.h, C.int32_t(job), C.int32_t(node)))
}
func (g *GpuRound) ClearAllocated() error { return g.check(C.asched_clear_allocated(g.h)) } // :1178

// GetNode (:358-394): AllocatableByPriority [P][R] and the jobs on the node with their evicted flag.
func (g *GpuRound) GetNode(node int32, P int) ([]int64, []int32, []uint8, error) {
	alloc := make([]int64, P*len(g.resNames))
	if err := g.check(C.asched_get_alloc(g.h, C.int32_t(node), i64p(alloc))); err != nil {
		return nil, nil, nil, err
	}
	jobs, ev := make([]int32, len(g.jobs)+1), make([]uint8, len(g.jobs)+1)
	n := C.asched_get_node_jobs(g.h, C.int32_t(node), i32p(jobs), u8p(ev), nil, C.int32_t(len(jobs)))
	if n < 0 {
		return nil, nil, nil, g.check(n)
	}
	return alloc, jobs[:int(n)], ev[:int(n)], nil
}

// IndexedNodeLabelValues (:340-343) as interned ids; ok == false when the label is not indexed.
func (g *GpuRound) IndexedNodeLabelValues(label string) ([]int32, bool) {
	out := make([]int32, len(g.nodes)+1)
	n := C.asched_indexed_node_label_values(g.h, C.int32_t(g.strs.id(label)), i32p(out), C.int32_t(len(out)))
	if n < 0 {
		return nil, false
	}
	return out[:int(n)], true
}

// SubmitCheck: SubmitChecker.getSchedulingResult's per-pool core for a batch (submitcheck.go:342-371); unit u = jobs[off[u]:off[u+1]].
func (g *GpuRound) SubmitCheck(off, jobs, flags []int32) ([]C.asched_submit_result, error) {
	res := make([]C.asched_submit_result, len(off)-1)
	if len(res) == 0 {
		return res, nil
	}
	rc := C.asched_submit_check(g.h, C.int32_t(len(res)), i32p(off), i32p(jobs), i32p(flags), &res[0])
	return res, g.check(rc)
}

// ---- one pool on several GPUs: the handle's communicator (include/armada_sched.h "The communicator"; no reference counterpart — FairSchedulingAlgo runs a pool on
// one goroutine, scheduling_algo.go:165).  The library enqueues ncclAllReduce itself on the handle's stream; the Go side only moves the 128-byte unique id between
// the scheduler replicas (one process — or one locked OS thread — per GPU) through whatever it already has: the leader's gRPC, a Pulsar message, a shared file.

// ExcludedNodes is PodSchedulingContext.NumExcludedNodesByReason (scheduling/context/pod.go:51) for a job whose last node selection ended without a node:
// asched_excluded_nodes returns what the reference's reason strings are made of (interned ids, a resource column and two raw quantities) and the strings are
// built here with the reference's OWN reason types (nodedb/nodematching.go:14-125), so a report prints what the CPU scheduler would print.  nil, nil = nothing on
// record (the job got a node, was never attempted, or failed a constraint before any node was looked at).  For the first untolerated taint of a node type to be
// the reference's (node_type.go:87-97 sorts taints by key STRING) UploadNodes interns the taint keys in lexicographic order before anything else.
func (g *GpuRound) ExcludedNodes(job int32) (map[string]int, error) {
	buf := make([]C.asched_excluded_reason, 64)
	n := C.asched_excluded_nodes(g.h, C.int32_t(job), &buf[0], C.int32_t(len(buf)))
	if int(n) > len(buf) {
		buf = make([]C.asched_excluded_reason, int(n))
		n = C.asched_excluded_nodes(g.h, C.int32_t(job), &buf[0], C.int32_t(len(buf)))
	}
	if n < 0 {
		return nil, g.check(n)
	}
	if n == 0 {
		return nil, nil
	}
	quantity := func(col C.int32_t, raw C.int64_t) k8sResource.Quantity { // ResourceList.asQuantity (resource_list.go:292-301): raw * 10^scale of the column
		scale, _ := g.rlf.GetScale(g.resNames[col])
		return *k8sResource.NewScaledQuantity(int64(raw), scale)
	}
	out := make(map[string]int, int(n))
	for _, r := range buf[:n] {
		var s string
		switch r.kind {
		case C.ASCHED_EXCL_IMPLICIT:
			s = nodedb.PodRequirementsNotMetReasonInsufficientResources
		case C.ASCHED_EXCL_UNTOLERATED_TAINT:
			t := v1.Taint{Key: g.strs.str(int32(r.a)), Value: g.strs.str(int32(r.b)), Effect: effectName(int32(r.c))}
			if r.a == -2 { // the taint an unschedulable node carries (internaltypes/unschedulable.go:12-18, node.go:127-129)
				t = internaltypes.UnschedulableTaint()
			}
			s = (&nodedb.UntoleratedTaint{Taint: t}).String()
		case C.ASCHED_EXCL_MISSING_LABEL:
			s = (&nodedb.MissingLabel{Label: g.strs.str(int32(r.a))}).String()
		case C.ASCHED_EXCL_UNMATCHED_LABEL:
			s = (&nodedb.UnmatchedLabel{Label: g.strs.str(int32(r.a)), PodValue: g.strs.str(int32(r.b)), NodeValue: g.strs.str(int32(r.c))}).String()
		case C.ASCHED_EXCL_UNMATCHED_AFFINITY:
			s = (&nodedb.UnmatchedNodeSelector{NodeSelector: g.jobs[job].PodRequirements().GetAffinityNodeSelector()}).String()
		case C.ASCHED_EXCL_INSUFFICIENT_RESOURCES:
			s = (&nodedb.InsufficientResources{ResourceName: g.resNames[r.a], Required: quantity(r.a, r.required), Available: quantity(r.a, r.available)}).String()
		case C.ASCHED_EXCL_DISALLOWED_RESOURCE:
			s = "job requests disallowed resource and therefore cannot be scheduled" // nodedb.disallowedResourceRequested (nodedb.go:27, unexported)
		}
		out[s] += int(r.count)
	}
	return out, nil
}

// SetExcludedNodes bounds how many failed node selections a round keeps a record of (default 1 024; 0 turns the records — and their one wide pass per failed attempt — off).
func (g *GpuRound) SetExcludedNodes(maxFailedSelections int) error {
	return g.check(C.asched_set_excluded_nodes(g.h, C.int32_t(maxFailedSelections)))
}

func effectName(e int32) v1.TaintEffect {
	switch e {
	case C.ASCHED_EFFECT_NO_SCHEDULE:
		return v1.TaintEffectNoSchedule
	case C.ASCHED_EFFECT_PREFER_NO_SCHEDULE:
		return v1.TaintEffectPreferNoSchedule
	case C.ASCHED_EFFECT_NO_EXECUTE:
		return v1.TaintEffectNoExecute
	}
	return ""
}

// CommUniqueId is called by ONE rank (ncclGetUniqueId); every rank then passes the same bytes to CommInit.
func CommUniqueId() ([128]byte, error) {
	var id C.asched_unique_id
	var out [128]byte
	if rc := C.asched_comm_unique_id(&id); rc != 0 {
		return out, errors.Errorf("asched_comm_unique_id: RCCL is not available (rc %d)", int(rc))
	}
	copy(out[:], C.GoBytes(unsafe.Pointer(&id.bytes[0]), 128))
	return out, nil
}

// CommInit == ncclCommInitRank on this handle's GPU; blocks until all `world` ranks have called it.
func (g *GpuRound) CommInit(id [128]byte, rank, world int) error {
	var cid C.asched_unique_id
	for i := range id {
		cid.bytes[i] = C.char(id[i])
	}
	return g.check(C.asched_comm_init(g.h, &cid, C.int32_t(rank), C.int32_t(world)))
}

func (g *GpuRound) CommDestroy() error { return g.check(C.asched_comm_destroy(g.h)) }

// ShardRound: this handle is one of the communicator's replicas of ONE pool (same nodes, jobs and queues on every replica).  From now on its rounds split their wide passes
// over the nodes across the replicas and all-reduce each pass's two result words (asched_shard_round; INTEGRATION.md 3a): every replica's Schedule returns the same round,
// the reference's.  Call after CommInit; ShardRound(false) goes back to whole passes.
func (g *GpuRound) ShardRound(on bool) error {
	v := C.int32_t(0)
	if on {
		v = 1
	}
	return g.check(C.asched_shard_round(g.h, v))
}

// ShardArea / ShardOpen / ShardPeers: the GPU-to-GPU exchange of sharded passes (INTEGRATION.md 3a): this replica's exchange area and its IPC handle bytes; another
// process' area mapped into this one; the list of every replica's area (areas[rank] = the own one) — from then on the handle's rounds split their wide passes.
func (g *GpuRound) ShardArea() (unsafe.Pointer, [64]byte, error) {
	var a C.asched_shard_area_t
	var ipc [64]byte
	if err := g.check(C.asched_shard_area(g.h, &a)); err != nil {
		return nil, ipc, err
	}
	for i := range ipc {
		ipc[i] = byte(a.ipc[i])
	}
	return a.ptr, ipc, nil
}

func (g *GpuRound) ShardOpen(ipc [64]byte) (unsafe.Pointer, error) {
	var out unsafe.Pointer
	err := g.check(C.asched_shard_open(g.h, (*C.char)(unsafe.Pointer(&ipc[0])), &out))
	return out, err
}

func (g *GpuRound) ShardPeers(areas []unsafe.Pointer, rank int) error {
	if len(areas) == 0 {
		return g.check(C.asched_shard_peers(g.h, nil, 0, 0))
	}
	return g.check(C.asched_shard_peers(g.h, (*unsafe.Pointer)(unsafe.Pointer(&areas[0])), C.int32_t(len(areas)), C.int32_t(rank)))
}

// ErrPeer: a collective entry point returned ASCHED_ERR_PEER — another rank of the communicator failed in front of the exchange (its own call returns the cause); nothing
// was exchanged and the communicator stays usable.  (Every rank all-reduces one status word before the data: a rank-local failure can no longer leave the others in RCCL.)
func (g *GpuRound) ErrPeer(err error) bool {
	return err != nil && g.lastCode == C.ASCHED_ERR_PEER
}

// FitSelectBatchSharded: this handle holds rows [rankOffset, rankOffset+len(nodes)) of the pool's nodes in index order; every rank calls it with the same jobs.
// EXACT: out[i] = position of the chosen node among ALL nodes of the pool in index order (-1: none) == nodeDb.SelectNodeForJobWithTxn's first fit at `priority`
// on a NodeDb holding all shards (nodedb.go:840-879).  fieldBits[k] = bits of floor(max allocatable / resolution) of indexed resource k over ALL shards.
func (g *GpuRound) FitSelectBatchSharded(jobs []int32, priority int32, fieldBits []int32, rankBits int32, rankOffset int64) ([]int32, error) {
	var lay C.asched_global_key_layout
	lay.n_fields = C.int32_t(len(fieldBits))
	for i, b := range fieldBits {
		lay.field_bits[i] = C.int32_t(b)
	}
	lay.rank_bits = C.int32_t(rankBits)
	lay.rank_offset = C.int64_t(rankOffset)
	out := make([]int32, len(jobs))
	if err := g.check(C.asched_fit_select_batch_sharded(g.h, C.int32_t(len(jobs)), i32p(jobs), C.int32_t(priority), &lay, i32p(out))); err != nil {
		return nil, err
	}
	return out, nil
}

// RoundExchange is the queue-hash mode of BASELINE's north_star after Schedule on every rank's replica: APPROXIMATE (DESIGN.md 7 — a conflict set is replayed, the
// result is feasible but not the reference's); per job: node after the accepted placements and all preemptions, priority there, replay flag.
func (g *GpuRound) RoundExchange(numJobs int) (C.asched_delta_summary, []int32, []int32, []uint8, error) {
	var sum C.asched_delta_summary
	node, prio, replay := make([]int32, numJobs), make([]int32, numJobs), make([]uint8, numJobs)
	err := g.check(C.asched_round_exchange(g.h, &sum, i32p(node), i32p(prio), u8p(replay)))
	return sum, node, prio, replay, err
}
