// mock_rccl.cpp -- TEST INFRASTRUCTURE, not product code: an in-process stand-in for librccl that
// lets N ranks of libgfship's own RCCL transport (csrc/transport.hip) run as N threads of ONE
// process on ONE GPU (the real RCCL refuses two ranks on one device, and this work has one GPU).
// libgfship opens it instead of librccl.so.1 when GFSHIP_RCCL_LIBRARY names it (rccl_load).
//
// Semantics kept from NCCL: point-to-point operations are matched per (sender, receiver) pair in
// posting order -- the k-th ncclSend from A to B meets the k-th ncclRecv from A on B -- and their
// counts must agree (a mismatch is recorded and reported as an error: mock_rccl_mismatches); a group
// posts all its sends before it waits for anything, so groups of sends and receives between
// neighbours cannot deadlock; ncclAllGather gathers in rank order.  Everything is synchronous on the
// host (stream synchronised before a buffer is published and after it has been filled): slow, and
// correct by construction.  A wrong peer[], a wrong rank_of, a left / right mix-up or a posting-order
// mismatch between two distinct peers therefore shows up as wrong bits (or as a count mismatch, or as
// the 60 s time-out of a wait that can never be served) in tests/test_gpu_mock_rccl.py.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Msg {
  const void * buf;
  size_t count;
  bool consumed = false;
};

struct World {
  int nranks = 0;
  std::mutex m;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<Msg *>> box;     // (src, dst) -> posted sends, in order
  std::vector<const void *> ag_send;
  int ag_arrived = 0, ag_done = 0;
  unsigned long long ag_gen = 0, ag_gen2 = 0;
};

struct Handle {
  World * w;
  int rank;
};

struct Op {
  bool send;
  void * buf;
  size_t count;
  int peer;
  Handle * h;
  hipStream_t stream;
};

std::mutex g_m;
std::map<std::string, World *> g_worlds;
unsigned long long g_ids = 0;
unsigned long long g_mismatches = 0, g_messages = 0, g_timeouts = 0;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

const auto WAIT = std::chrono::seconds (60);

ncclResult_t run_ops (std::vector<Op> & ops)
{
  if (ops.empty ()) return ncclSuccess;
  // what is sent is complete
  for (const Op & o : ops)
    if (o.send && hipStreamSynchronize (o.stream) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<Msg *> mine;
  for (const Op & o : ops)
    if (o.send) {
      World * w = o.h->w;
      Msg * msg = new Msg { o.buf, o.count };
      {
	std::lock_guard<std::mutex> lk (w->m);
	w->box[{ o.h->rank, o.peer }].push_back (msg);
	g_messages++;
      }
      w->cv.notify_all ();
      mine.push_back (msg);
    }
  ncclResult_t res = ncclSuccess;
  for (const Op & o : ops)
    if (!o.send) {
      World * w = o.h->w;
      Msg * msg = nullptr;
      {
	std::unique_lock<std::mutex> lk (w->m);
	auto & q = w->box[{ o.peer, o.h->rank }];
	if (!w->cv.wait_for (lk, WAIT, [&] { return !q.empty (); })) {
	  fprintf (stderr, "mock_rccl: rank %d: no send from rank %d arrived for a receive of %zu\n",
		   o.h->rank, o.peer, o.count);
	  g_timeouts++;
	  res = ncclInternalError;
	  continue;
	}
	msg = q.front ();
	q.pop_front ();
      }
      if (msg->count != o.count) {
	fprintf (stderr, "mock_rccl: rank %d <- %d: send of %zu meets receive of %zu\n", o.h->rank, o.peer,
		 msg->count, o.count);
	std::lock_guard<std::mutex> lk (w->m);
	g_mismatches++;
	res = ncclInvalidArgument;
      }
      else if (hipMemcpyAsync (o.buf, msg->buf, o.count*sizeof (double), hipMemcpyDeviceToDevice,
			       o.stream) != hipSuccess ||
	       hipStreamSynchronize (o.stream) != hipSuccess)
	res = ncclUnhandledCudaError;
      {
	std::lock_guard<std::mutex> lk (w->m);
	msg->consumed = true;
      }
      w->cv.notify_all ();
    }
  // my send buffers may be reused once they have been read
  for (size_t q = 0, k = 0; q < ops.size (); q++)
    if (ops[q].send) {
      World * w = ops[q].h->w;
      Msg * msg = mine[k++];
      std::unique_lock<std::mutex> lk (w->m);
      if (!w->cv.wait_for (lk, WAIT, [&] { return msg->consumed; })) {
	fprintf (stderr, "mock_rccl: rank %d: the send of %zu to rank %d was never received\n",
		 ops[q].h->rank, msg->count, ops[q].peer);
	g_timeouts++;
	res = ncclInternalError;
	continue;       /* the message stays in the mailbox: leaked, the test has failed anyway */
      }
      lk.unlock ();
      delete msg;
    }
  return res;
}

} // namespace

extern "C" {

unsigned long long mock_rccl_mismatches (void) { return g_mismatches; }
unsigned long long mock_rccl_messages (void) { return g_messages; }
unsigned long long mock_rccl_timeouts (void) { return g_timeouts; }

ncclResult_t ncclGetUniqueId (ncclUniqueId * id)
{
  std::lock_guard<std::mutex> lk (g_m);
  memset (id, 0, sizeof (*id));
  snprintf (id->internal, sizeof (id->internal), "mock-rccl-world-%llu", ++g_ids);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank (ncclComm_t * comm, int nranks, ncclUniqueId id, int rank)
{
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  std::lock_guard<std::mutex> lk (g_m);
  std::string key (id.internal, strnlen (id.internal, sizeof (id.internal)));
  World *& w = g_worlds[key];
  if (!w) {
    w = new World;
    w->nranks = nranks;
    w->ag_send.assign (nranks, nullptr);
  }
  if (w->nranks != nranks) return ncclInvalidArgument;
  *comm = (ncclComm_t) new Handle { w, rank };
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy (ncclComm_t comm)
{
  delete (Handle *) comm;
  return ncclSuccess;
}

ncclResult_t ncclCommCount (const ncclComm_t comm, int * count)
{
  *count = ((Handle *) comm)->w->nranks;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart ()
{
  t_depth++;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd ()
{
  if (--t_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap (t_ops);
  return run_ops (ops);
}

ncclResult_t ncclSend (const void * buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm,
		       hipStream_t stream)
{
  Handle * h = (Handle *) comm;
  if (type != ncclDouble || peer < 0 || peer >= h->w->nranks) return ncclInvalidArgument;
  t_ops.push_back (Op { true, const_cast<void *> (buf), count, peer, h, stream });
  if (t_depth == 0) {
    std::vector<Op> ops;
    ops.swap (t_ops);
    return run_ops (ops);
  }
  return ncclSuccess;
}

ncclResult_t ncclRecv (void * buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm,
		       hipStream_t stream)
{
  Handle * h = (Handle *) comm;
  if (type != ncclDouble || peer < 0 || peer >= h->w->nranks) return ncclInvalidArgument;
  t_ops.push_back (Op { false, buf, count, peer, h, stream });
  if (t_depth == 0) {
    std::vector<Op> ops;
    ops.swap (t_ops);
    return run_ops (ops);
  }
  return ncclSuccess;
}

ncclResult_t ncclAllGather (const void * send, void * recv, size_t count, ncclDataType_t type,
			    ncclComm_t comm, hipStream_t stream)
{
  Handle * h = (Handle *) comm;
  World * w = h->w;
  if (type != ncclDouble) return ncclInvalidArgument;
  if (hipStreamSynchronize (stream) != hipSuccess) return ncclUnhandledCudaError;
  {
    std::unique_lock<std::mutex> lk (w->m);
    w->ag_send[h->rank] = send;
    const unsigned long long gen = w->ag_gen;
    if (++w->ag_arrived == w->nranks) { w->ag_arrived = 0; w->ag_gen++; w->cv.notify_all (); }
    else if (!w->cv.wait_for (lk, WAIT, [&] { return w->ag_gen != gen; })) { g_timeouts++; return ncclInternalError; }
  }
  ncclResult_t res = ncclSuccess;
  for (int r = 0; r < w->nranks; r++)
    if (hipMemcpyAsync ((double *) recv + (size_t) r*count, w->ag_send[r], count*sizeof (double),
			hipMemcpyDeviceToDevice, stream) != hipSuccess)
      res = ncclUnhandledCudaError;
  if (hipStreamSynchronize (stream) != hipSuccess) res = ncclUnhandledCudaError;
  {
    std::unique_lock<std::mutex> lk (w->m);
    const unsigned long long gen = w->ag_gen2;
    if (++w->ag_done == w->nranks) { w->ag_done = 0; w->ag_gen2++; w->cv.notify_all (); }
    else if (!w->cv.wait_for (lk, WAIT, [&] { return w->ag_gen2 != gen; })) { g_timeouts++; return ncclInternalError; }
  }
  return res;
}

const char * ncclGetErrorString (ncclResult_t r)
{
  switch (r) {
  case ncclSuccess: return "success";
  case ncclInvalidArgument: return "mock: invalid argument / count mismatch";
  case ncclInternalError: return "mock: a wait timed out";
  case ncclUnhandledCudaError: return "mock: HIP error";
  default: return "mock: error";
  }
}

} // extern "C"
