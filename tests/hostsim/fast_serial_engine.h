// The node engine of the serial build (tests/hostsim; see fast_serial.h): TEST INFRASTRUCTURE, included by round_fast.h only under ASCHED_HOSTSIM.
// serial build: the engine runs at post time; the control code still proceeds on the assumption that the job fits and takes the
// iteration back at the next settle point when it did not — the same control flow as on the device
static Dev* g_hsEngPending = nullptr;   // a posted job the serial engine has not served yet (lag mode)
DEV void engineStart(Dev&, FastS& S) { g_hsEngPending = nullptr; g_engS = S; g_engS.statScanSteps = 0; g_engS.engSeq = 0; FL.eng.cancel = 0; FL.eng.live = 1; }
DEV void engineStop(Dev&, FastS& S) { FL.eng.live = 0; S.statScanSteps += g_engS.statScanSteps; if (g_engS.statL0Max > S.statL0Max) S.statL0Max = g_engS.statL0Max; }
DEV void enginePost(Dev& d, KREF k, FastS& S, int job, int q, int pc, int32_t prio, int32_t cutoff, int nl) {
  IterBackup& b = FL.bk;
  b.hot = FL.hot[q];
  b.kA = FL.kA[q]; b.kX = FL.kX[q]; b.kY = FL.kY[q]; b.effA = FL.effA[q]; b.effX = FL.effX[q]; b.effY = FL.effY[q];
  b.globalTokens = S.globalTokens; b.pc = pc; b.inHeap = FL.inHeap[q];
  FL.eng.tail = FL.headTail[q]; memcpy(FL.eng.req, FL.headReq[q], sizeof FL.eng.req);
  FL.eng.job = job; FL.eng.prio = prio; FL.eng.cutoff = cutoff; FL.eng.nl = nl;
  S.engSeq++;
  // On the device the engine wave serves the job WHILE the control wave goes on with the queue side of the next iteration; the serial build runs it here, at post time —
  // or, in lag mode (HS_RING_LAG, fast_serial.h), only when the control code asks for the verdict (engineWait): the other extreme of the same race.  Both orders are
  // interleavings the device can produce, so both must give the oracle's round.
  if (hsLagSeed() && (hsLagRand() & 1)) { g_hsEngPending = &d; return; }
  FL.eng.status = engineServe(d, k, g_engS);
}
DEV int engineWait(const FastS&) {
  if (g_hsEngPending) { Dev& d = *g_hsEngPending; g_hsEngPending = nullptr; const FastK k = fastKRef(d); FL.eng.status = engineServe(d, k, g_engS); }
  return FL.eng.status;
}
