// mhh_host_rccl.h -- C++ host driver of the slab-decomposed path (npx = 1, npy = N, one process per GPU): the exchanges the
// reference's CPU-MPI build issues through MPI, issued here through RCCL (point-to-point xGMI links between the GPUs of a node):
//
//   Master_rccl             rank layout + communicator + Master::max           src/master_parallel.cxx:103-162, 233-266
//   Boundary_cyclic_slab    east-west wrap on the device, north-south rows     src/boundary_cyclic.cxx:116-176
//                           to the ring neighbours (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd)
//   Transpose               exec_xy / exec_yx of the spectral pressure as one  src/transpose.cxx:170-219
//                           grouped all-to-all (every pair of ranks exchanges one block: all xGMI links busy at once)
//   Pres_slab               Pres_2::exec around the two transposes             src/pres_2.cxx:66-94, src/fft.cxx:451-583
//
// The library (include/mhh_hip.h) packs, transforms, solves and unpacks; this header only owns the message buffers and issues
// the collectives on the caller's stream, so that kernels and messages are ordered by the stream alone (no host synchronisation
// inside a sub-step). Unlike mhh_host.h it needs the HIP runtime and RCCL headers, hence a file of its own.
// tests/cpp/host_step.cpp drives it with a one-rank communicator on the one-GPU test box (every message goes to self through
// RCCL); the N > 1 instantiation is the same code with the ids of N processes.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include "mhh_host.h"

namespace mhh_host
{
inline void hip_check(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }
inline void nccl_check(ncclResult_t r, const char* what) { if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r)); }

// The slice of Master the hot path uses. One process per GPU; `mpicoordy` = rank, `npy` = number of ranks (npx = 1).
class Master_rccl
{
    public:
        Master_rccl() = default;
        Master_rccl(const Master_rccl&) = delete;
        ~Master_rccl() { if (comm) ncclCommDestroy(comm); }
        static ncclUniqueId unique_id() { ncclUniqueId id; nccl_check(ncclGetUniqueId(&id), "ncclGetUniqueId"); return id; }
        // every rank calls this with the id rank 0 made (handed over out of band, as MPI_Init does for the reference)
        void init(int nranks, int rank, const ncclUniqueId& id, hipStream_t s = nullptr)
        {
            npy = nranks; mpicoordy = rank; stream = s;
            nccl_check(ncclCommInitRank(&comm, nranks, id, rank), "ncclCommInitRank");
            hip_check(hipMalloc(&scalar, sizeof(double)), "hipMalloc");
        }
        int south() const { return (mpicoordy + npy - 1) % npy; }
        int north() const { return (mpicoordy + 1) % npy; }
        // Master::max(double*, 1) (src/master_parallel.cxx:233-266): the one scalar reduction of the hot path (cfl, dnmul, divergence)
        double max(double v)
        {
            hip_check(hipMemcpyAsync(scalar, &v, sizeof(double), hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
            nccl_check(ncclAllReduce(scalar, scalar, 1, ncclDouble, ncclMax, comm, stream), "ncclAllReduce");
            hip_check(hipMemcpyAsync(&v, scalar, sizeof(double), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
            hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
            return v;
        }
        ncclComm_t comm = nullptr;
        int npy = 1, mpicoordy = 0;
        hipStream_t stream = nullptr;
    private:
        double* scalar = nullptr;
};

// Boundary_cyclic for a y-slab: exec_g(list) fills the ghost cells of up to 8 fields with ONE message pair per neighbour.
// rows_south / rows_north < jgc exchange only what the next kernel reads (pres_2 input: vt[j+1]; output: p[j-1]).
template<typename TF>
class Boundary_cyclic_slab
{
    public:
        Boundary_cyclic_slab(Master_rccl& m, Grid<TF>& g) : master(m), grid(g) {}
        ~Boundary_cyclic_slab() { if (send) (void)hipFree(send); if (recv) (void)hipFree(recv); }
        void exec_g(const std::vector<TF*>& fields, int rows_south = -1, int rows_north = -1)
        {
            const auto& gd = grid.get_grid_data();
            const int rs = rows_south < 0 ? gd.jgc : rows_south, rn = rows_north < 0 ? gd.jgc : rows_north;
            if (rs == gd.jgc && rn == gd.jgc) exec_ew(fields);      // a full exchange also refreshes the east-west ghosts (all j, ghost rows included)
            exec_ns(fields, rs, rn, master.stream);
        }
        // the east-west wrap on the device (no message)
        void exec_ew(const std::vector<TF*>& fields)
        {
            mhh_grid g = grid.abi();
            std::vector<void*> ptrs(fields.begin(), fields.end());
            mhh_check(mhh_boundary_cyclic_n(&g, ptrs.data(), (int)fields.size(), MHH_EDGE_EW, master.stream));
        }
        // the north-south rows: pack, one message pair per neighbour, unpack -- all on `stream` (the overlapped sub-step puts them on
        // an exchange stream of its own while the rows that need no halo are worked, src/boundary_cyclic.cxx:116-176)
        void exec_ns(const std::vector<TF*>& fields, int rs, int rn, hipStream_t stream)
        {
            const auto& gd = grid.get_grid_data();
            mhh_grid g = grid.abi();
            const int nf = (int)fields.size();
            std::vector<void*> ptrs(fields.begin(), fields.end());
            void* st = stream;
            const size_t per_row = (size_t)nf * gd.kcells * gd.icells, nn = rn * per_row, ns = rs * per_row;
            reserve((nn + ns) * sizeof(TF));
            // one send and one receive buffer, [northbound | southbound]
            TF* s_north = static_cast<TF*>(send); TF* s_south = s_north + nn;
            TF* r_south = static_cast<TF*>(recv); TF* r_north = r_south + nn;     // what the south neighbour sent north | what the north one sent south
            mhh_check(mhh_halo_pack_rows(&g, ptrs.data(), nf, rs, rn, s_south, s_north, st));
            nccl_check(ncclGroupStart(), "ncclGroupStart");
            if (nn) { nccl_check(ncclSend(s_north, nn*sizeof(TF), ncclChar, master.north(), master.comm, stream), "ncclSend");
                      nccl_check(ncclRecv(r_south, nn*sizeof(TF), ncclChar, master.south(), master.comm, stream), "ncclRecv"); }
            if (ns) { nccl_check(ncclSend(s_south, ns*sizeof(TF), ncclChar, master.south(), master.comm, stream), "ncclSend");
                      nccl_check(ncclRecv(r_north, ns*sizeof(TF), ncclChar, master.north(), master.comm, stream), "ncclRecv"); }
            nccl_check(ncclGroupEnd(), "ncclGroupEnd");
            mhh_check(mhh_halo_unpack_rows(&g, ptrs.data(), nf, rs, rn, r_south, r_north, st));
        }
        void exec_g(TF* data) { exec_g(std::vector<TF*>{data}); }
    private:
        void reserve(size_t bytes)
        {
            if (bytes <= cap) return;
            if (send) (void)hipFree(send);
            if (recv) (void)hipFree(recv);
            hip_check(hipMalloc(&send, bytes), "hipMalloc"); hip_check(hipMalloc(&recv, bytes), "hipMalloc");
            cap = bytes;
        }
        Master_rccl& master; Grid<TF>& grid;
        void* send = nullptr; void* recv = nullptr; size_t cap = 0;
};

// Transpose::exec_xy / exec_yx of the packed spectral field: equal blocks [peer][...] to and from every rank
class Transpose
{
    public:
        explicit Transpose(Master_rccl& m) : master(m) {}
        void exec(const void* sendbuf, void* recvbuf, size_t bytes_per_peer) { exec(sendbuf, recvbuf, bytes_per_peer, master.stream); }
        void exec(const void* sendbuf, void* recvbuf, size_t bytes_per_peer, hipStream_t stream)
        {
            nccl_check(ncclGroupStart(), "ncclGroupStart");
            for (int r = 0; r < master.npy; ++r)
            {
                nccl_check(ncclSend(static_cast<const char*>(sendbuf) + (size_t)r*bytes_per_peer, bytes_per_peer, ncclChar, r, master.comm, stream), "ncclSend");
                nccl_check(ncclRecv(static_cast<char*>(recvbuf) + (size_t)r*bytes_per_peer, bytes_per_peer, ncclChar, r, master.comm, stream), "ncclRecv");
            }
            nccl_check(ncclGroupEnd(), "ncclGroupEnd");
        }
    private:
        Master_rccl& master;
};

// Pres_2 on a y-slab (the reference's Pres interface, include/pres.h:39-85)
template<typename TF>
class Pres_slab
{
    public:
        Pres_slab(Master_rccl& m, Grid<TF>& g, Fields<TF>& f) : master(m), grid(g), fields(f), halo(m, g), transpose(m) {}
        ~Pres_slab() { clear_device(); }
        void init() {}
        void set_values() {}
        void create(Stats&) {}
        void set_reduce_workspace(void* device_scratch) { work = device_scratch; }
        void prepare_device()
        {
            const auto& gd = grid.get_grid_data();
            mhh_grid g = grid.abi();
            mhh_check(mhh_pres_slab_plan_create(&g, gd.dz.data(), gd.dzhi.data(), fields.rhoref.data(), fields.rhorefh.data(), &plan));
            nbytes = (size_t)mhh_pres_slab_xbuf_elems(plan) * 2 * sizeof(TF);       // complex elements
            hip_check(hipMalloc(&xsend, nbytes), "hipMalloc"); hip_check(hipMalloc(&xrecv, nbytes), "hipMalloc");
        }
        // k-slices of the solve: the all-to-all of slice c travels on a second stream while slice c+1 is transformed (the reference's
        // FFT::exec_forward / Transpose::exec_xy work through batches of planes in turn as well, src/fft.cxx:451-583,
        // src/transpose.cxx:170-219). n = 1: whole transposes on the caller's stream. Call after prepare_device.
        void set_chunks(int n)
        {
            if (!plan) throw std::runtime_error("Pres_slab::set_chunks before prepare_device");
            const auto& gd = grid.get_grid_data();
            if (n < 1) n = 1;
            while (n > 1 && gd.ktot % n) --n;
            mhh_check(mhh_pres_slab_set_chunks(plan, n));
            nchunks = n;
            if (n > 1 && !comm_stream)
            {
                hip_check(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking), "hipStreamCreate");
                events.resize(4*(size_t)n);
                for (auto& e : events) hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
            }
            else if (n > 1 && events.size() < 4*(size_t)n)
            {
                const size_t old = events.size(); events.resize(4*(size_t)n);
                for (size_t m = old; m < events.size(); ++m) hip_check(hipEventCreateWithFlags(&events[m], hipEventDisableTiming), "hipEventCreate");
            }
        }
        int chunks() const { return nchunks; }
        void clear_device()
        {
            if (plan) { mhh_pres_slab_plan_destroy(plan); plan = nullptr; }
            if (xsend) { (void)hipFree(xsend); xsend = nullptr; }
            if (xrecv) { (void)hipFree(xrecv); xrecv = nullptr; }
            for (auto& e : events) (void)hipEventDestroy(e);
            events.clear();
            if (comm_stream) { (void)hipStreamDestroy(comm_stream); comm_stream = nullptr; }
            nchunks = 1;
        }
        void exec(double dt, Stats&)
        {
            if (!plan) throw std::runtime_error("Pres_slab::exec before prepare_device");
            mhh_grid g = grid.abi();
            mhh_fields f = abi_fields(fields);
            void* st = master.stream;
            halo.exec_g({fields.mt.at("v")->fld_g}, 1, 0);                 // pres_2 input reads vt[j+1] only (src/pres_2.cxx:181,193)
            // x stages with the transforms in LDS where the plan has them: input + x transform write the send buffer, x transform +
            // p + output read the receive buffer (mhh_pres_slab_lds_fwd / _bwd); otherwise the staged kernels around a packed array
            const bool lds = mhh_pres_slab_has_lds(plan) == 1;
            void* packed = mhh_pres_slab_packed(plan);
            if (!lds) mhh_check(mhh_pres_input_packed(&g, 2, &f, dt, packed, st));
            if (nchunks == 1)
            {
                if (lds) mhh_check(mhh_pres_slab_lds_fwd(plan, &g, &f, dt, xsend, 0, st));
                else     mhh_check(mhh_pres_fwd_x_pack(plan, &g, packed, xsend, st));
                transpose.exec(xsend, xrecv, nbytes / master.npy);            // Transpose::exec_xy
                if (lds)
                {
                    mhh_check(mhh_pres_slab_lds_fwd_y(plan, &g, xrecv, 0, st));
                    mhh_check(mhh_pres_solve_y(plan, &g, st));
                    mhh_check(mhh_pres_slab_lds_bwd_y(plan, &g, xsend, 0, st));
                }
                else mhh_check(mhh_pres_fwd_y_solve_bwd_y(plan, &g, xrecv, xsend, st));
                transpose.exec(xsend, xrecv, nbytes / master.npy);            // Transpose::exec_yx
                if (lds) mhh_check(mhh_pres_slab_lds_bwd(plan, &g, xrecv, &f, 0, st));
                else     mhh_check(mhh_pres_bwd_x_unpack_output(plan, &g, xrecv, &f, st));
            }
            else
            {
                const size_t seg = nbytes / nchunks;                       // slice c = bytes [c*seg, (c+1)*seg) of both buffers: an equal-split all-to-all of its own
                auto exchange = [&](int c, hipEvent_t ready, hipEvent_t done)
                {
                    hip_check(hipEventRecord(ready, master.stream), "hipEventRecord");
                    hip_check(hipStreamWaitEvent(comm_stream, ready, 0), "hipStreamWaitEvent");
                    transpose.exec(static_cast<char*>(xsend) + c*seg, static_cast<char*>(xrecv) + c*seg, seg / master.npy, comm_stream);
                    hip_check(hipEventRecord(done, comm_stream), "hipEventRecord");
                };
                const int n = nchunks;
                for (int c = 0; c < n; ++c)
                {
                    if (lds) mhh_check(mhh_pres_slab_lds_fwd(plan, &g, &f, dt, xsend, c, st));
                    else     mhh_check(mhh_pres_fwd_x_pack_chunk(plan, &g, packed, xsend, c, st));
                    exchange(c, events[c], events[n + c]);
                }
                for (int c = 0; c < n; ++c)
                {
                    hip_check(hipStreamWaitEvent(master.stream, events[n + c], 0), "hipStreamWaitEvent");
                    if (lds) mhh_check(mhh_pres_slab_lds_fwd_y(plan, &g, xrecv, c, st));
                    else     mhh_check(mhh_pres_fwd_y_chunk(plan, &g, xrecv, c, st));
                }
                mhh_check(mhh_pres_solve_y(plan, &g, st));
                for (int c = 0; c < n; ++c)
                {
                    if (lds) mhh_check(mhh_pres_slab_lds_bwd_y(plan, &g, xsend, c, st));
                    else     mhh_check(mhh_pres_bwd_y_chunk(plan, &g, xsend, c, st));
                    exchange(c, events[2*n + c], events[3*n + c]);
                }
                for (int c = 0; c < n; ++c)
                {
                    hip_check(hipStreamWaitEvent(master.stream, events[3*n + c], 0), "hipStreamWaitEvent");
                    if (lds) mhh_check(mhh_pres_slab_lds_bwd(plan, &g, xrecv, &f, c, st));
                    else     mhh_check(mhh_pres_bwd_x_chunk(plan, &g, xrecv, c, st));
                }
                if (!lds) mhh_check(mhh_pres_unpack_output_slab(plan, &g, &f, st));
            }
            halo.exec_g({fields.sd.at("p")->fld_g}, 0, 1);                 // output of the southernmost row reads p[j-1] (:383-385)
            mhh_check(mhh_pres_output_south_row(&g, &f, st));
        }
        TF check_divergence()
        {
            mhh_grid g = grid.abi();
            mhh_fields f = abi_fields(fields);
            double div = 0;
            mhh_check(mhh_pres_check_divergence(&g, 2, &f, work, &div, master.stream));
            return static_cast<TF>(master.max(div));
        }
    private:
        Master_rccl& master; Grid<TF>& grid; Fields<TF>& fields;
        Boundary_cyclic_slab<TF> halo; Transpose transpose;
        mhh_pres_slab_plan* plan = nullptr; void* work = nullptr;
        void* xsend = nullptr; void* xrecv = nullptr; size_t nbytes = 0;
        int nchunks = 1; hipStream_t comm_stream = nullptr; std::vector<hipEvent_t> events;
};

// set_prognostic_cyclic_bcs + diff->exec_viscosity + advec->exec + diff->exec of one sub-step on a y-slab (src/model.cxx:346-392)
// with the north-south exchange of u, v, w and the scalar -- jgc rows of four fields each way, the largest message of the step --
// travelling on a stream of its own while the rows that read no north-south halo are worked: evisc on rows [jstart+1, jend-1), the
// tendencies on rows [jstart+4, jend-4); both edge strips follow in ONE launch per operator once the halos are in
// (mhh_diff_exec_viscosity_rows2, mhh_rhs_exec_rows2). evisc on the two ghost rows next to the slab is evaluated locally from the
// velocity halos (same operands as on the neighbour: same bits), so it is not exchanged. Row-wise calls give the bits of the
// whole-slab calls. Needs advec_2i5 + diff_smag2, one scalar, jgc >= 3 and jmax >= 12; can_overlap() tells.
template<typename TF>
class Substep_slab
{
    public:
        Substep_slab(Master_rccl& m, Grid<TF>& g, Fields<TF>& f, Boundary_cyclic_slab<TF>& h) : master(m), grid(g), fields(f), halo(h) {}
        ~Substep_slab()
        {
            if (ev_ready) (void)hipEventDestroy(ev_ready);
            if (ev_done) (void)hipEventDestroy(ev_done);
            if (comm_stream) (void)hipStreamDestroy(comm_stream);
        }
        bool can_overlap(const Advec<TF>& advec, const Diff<TF>& diff) const
        {
            const auto& gd = grid.get_grid_data();
            return advec.get_scheme() == MHH_ADVEC_2I5 && diff.get_scheme() == MHH_DIFF_SMAG2 && fields.sp.size() == 1 && gd.jgc >= 3 && gd.jmax >= 12;
        }
        void halo_visc_rhs(Advec<TF>& advec, Diff<TF>& diff, Thermo<TF>& thermo)
        {
            if (!can_overlap(advec, diff)) throw std::runtime_error("Substep_slab::halo_visc_rhs: scheme pair / slab too thin for the overlapped path");
            const auto& gd = grid.get_grid_data();
            if (!comm_stream)
            {
                hip_check(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking), "hipStreamCreate");
                hip_check(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming), "hipEventCreate");
                hip_check(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming), "hipEventCreate");
            }
            std::vector<TF*> prog{fields.mp.at("u")->fld_g, fields.mp.at("v")->fld_g, fields.mp.at("w")->fld_g};
            for (auto& it : fields.sp) prog.push_back(it.second->fld_g);
            halo.exec_ew(prog);
            hip_check(hipEventRecord(ev_ready, master.stream), "hipEventRecord");
            hip_check(hipStreamWaitEvent(comm_stream, ev_ready, 0), "hipStreamWaitEvent");
            halo.exec_ns(prog, gd.jgc, gd.jgc, comm_stream);
            hip_check(hipEventRecord(ev_done, comm_stream), "hipEventRecord");
            const int ja = gd.jstart + 4, jb = gd.jend - 4;
            diff.exec_viscosity_rows(thermo, gd.jstart + 1, gd.jend - 1, -1, -1, master.stream);
            diff.exec_with_advec_rows(advec, ja, jb, -1, -1, master.stream);
            hip_check(hipStreamWaitEvent(master.stream, ev_done, 0), "hipStreamWaitEvent");
            diff.exec_viscosity_rows(thermo, gd.jstart - 1, gd.jstart + 1, gd.jend - 1, gd.jend + 1, master.stream);
            diff.exec_with_advec_rows(advec, gd.jstart, ja, jb, gd.jend, master.stream);
        }
    private:
        Master_rccl& master; Grid<TF>& grid; Fields<TF>& fields; Boundary_cyclic_slab<TF>& halo;
        hipStream_t comm_stream = nullptr; hipEvent_t ev_ready = nullptr, ev_done = nullptr;
};
} // namespace mhh_host
