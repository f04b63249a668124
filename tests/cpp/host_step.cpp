// tests/cpp/host_step.cpp -- drives one sub-step of the hot path through the C++ host classes of
// microhh_amd/host/mhh_host.h (the reference's Advec/Diff/Pres/Boundary_cyclic interfaces over the C ABI), the way
// Model<TF>::exec does (src/model.cxx:346-411). Input and output are raw binary files written/read by
// tests/test_cpp_host.py, which checks the result against the CPU oracle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include "../../microhh_amd/host/mhh_host.h"
#include "../../microhh_amd/host/mhh_host_rccl.h"

using namespace mhh_host;
typedef double TF;

#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(2); } } while (0)

static std::vector<TF> rd(FILE* f, size_t n) { std::vector<TF> v(n); if (std::fread(v.data(), sizeof(TF), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(3); } return v; }
static TF* up(const std::vector<TF>& v) { TF* d; HIPCHK(hipMalloc(&d, v.size()*sizeof(TF))); HIPCHK(hipMemcpy(d, v.data(), v.size()*sizeof(TF), hipMemcpyHostToDevice)); return d; }
static void dn(FILE* f, const TF* d, size_t n) { std::vector<TF> v(n); HIPCHK(hipMemcpy(v.data(), d, n*sizeof(TF), hipMemcpyDeviceToHost)); std::fwrite(v.data(), sizeof(TF), n, f); }

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: host_step in.bin out.bin\n"); return 1; }
    FILE* in = std::fopen(argv[1], "rb"); if (!in) return 1;
    int hdr[8]; if (std::fread(hdr, sizeof(int), 8, in) != 8) return 1;
    double par[6]; if (std::fread(par, sizeof(double), 6, in) != 6) return 1;
    try
    {
        Grid<TF> grid; auto& gd = grid.gd;
        gd.itot = hdr[0]; gd.jtot = hdr[1]; gd.ktot = hdr[2]; gd.igc = hdr[3]; gd.jgc = hdr[4]; gd.kgc = hdr[5];
        const int sm = hdr[6];
        const bool fused = hdr[7] & 1, limited = hdr[7] & 2, buoy = hdr[7] & 4, slab = hdr[7] & 8, overlap = hdr[7] & 16, sliced = hdr[7] & 32;
        gd.imax = gd.itot; gd.jmax = gd.jtot; gd.kmax = gd.ktot;
        gd.icells = gd.itot + 2*gd.igc; gd.jcells = gd.jtot + 2*gd.jgc; gd.kcells = gd.ktot + 2*gd.kgc; gd.ijcells = gd.icells*gd.jcells; gd.ncells = gd.ijcells*gd.kcells;
        gd.istart = gd.igc; gd.jstart = gd.jgc; gd.kstart = gd.kgc; gd.iend = gd.istart + gd.itot; gd.jend = gd.jstart + gd.jtot; gd.kend = gd.kstart + gd.ktot;
        gd.xsize = par[0]; gd.ysize = par[1]; gd.zsize = par[2]; gd.dx = gd.xsize/gd.itot; gd.dy = gd.ysize/gd.jtot;
        const double dt = par[3];
        const size_t nk = gd.kcells, n3 = gd.ncells, n2 = gd.ijcells;
        gd.z = rd(in, nk); gd.zh = rd(in, nk); gd.dz = rd(in, nk); gd.dzh = rd(in, nk); gd.dzi = rd(in, nk); gd.dzhi = rd(in, nk); gd.dzi4 = rd(in, nk); gd.dzhi4 = rd(in, nk);
        gd.z_g = up(gd.z); gd.zh_g = up(gd.zh); gd.dz_g = up(gd.dz); gd.dzh_g = up(gd.dzh); gd.dzi_g = up(gd.dzi); gd.dzhi_g = up(gd.dzhi); gd.dzi4_g = up(gd.dzi4); gd.dzhi4_g = up(gd.dzhi4);

        Fields<TF> fields;
        fields.visc = par[4];
        fields.rhoref = rd(in, nk); fields.rhorefh = rd(in, nk);
        fields.rhoref_g = up(fields.rhoref); fields.rhorefh_g = up(fields.rhorefh);
        auto mk = [&](size_t n) { auto f = std::make_shared<Field3d<TF>>(); f->fld_g = up(rd(in, n)); return f; };
        for (const char* nm : {"u", "v", "w"}) fields.mp[nm] = mk(n3);
        fields.sp["th"] = mk(n3); fields.sp["th"]->visc = par[5];
        for (const char* nm : {"u", "v", "w"}) fields.mt[nm] = mk(n3);
        fields.st["th"] = mk(n3);
        fields.mp["u"]->flux_bot_g = up(rd(in, n2)); fields.mp["u"]->flux_top_g = up(rd(in, n2));
        fields.mp["v"]->flux_bot_g = up(rd(in, n2)); fields.mp["v"]->flux_top_g = up(rd(in, n2));
        fields.sp["th"]->flux_bot_g = up(rd(in, n2)); fields.sp["th"]->flux_top_g = up(rd(in, n2));
        Boundary<TF> boundary; boundary.swboundary = sm ? "surface" : "default";
        boundary.dudz_g = up(rd(in, n2)); boundary.dvdz_g = up(rd(in, n2)); boundary.dbdz_g = up(rd(in, n2));
        const std::vector<TF> z0m_h = rd(in, n2); boundary.z0m_g = up(z0m_h);
        Thermo<TF> thermo; thermo.swthermo = "dry"; thermo.thref_g = up(rd(in, nk)); thermo.grav = 9.81;
        if (buoy) thermo.threfh_g = up(rd(in, nk));
        std::fclose(in);
        for (const char* nm : {"evisc", "p"}) { auto f = std::make_shared<Field3d<TF>>(); f->fld_g = up(std::vector<TF>(n3, 0.)); fields.sd[nm] = f; }

        void* work; HIPCHK(hipMalloc(&work, mhh_reduce_work_bytes()));
        TF* mlen0; HIPCHK(hipMalloc((void**)&mlen0, nk*sizeof(TF)));
        Stats stats;
        Boundary_cyclic<TF> boundary_cyclic(grid);
        auto advec = Advec<TF>::factory(grid, fields, "2i5", 1.0, limited ? std::vector<std::string>{"th"} : std::vector<std::string>{});
        auto diff = Diff<TF>::factory(grid, fields, boundary, "smag2");
        auto pres = Pres<TF>::factory(grid, fields, "2");
        advec->set_reduce_workspace(work); diff->set_reduce_workspace(work); pres->set_reduce_workspace(work);
        diff->prepare_device(boundary, mlen0, [](void* d, const void* s, size_t n) { HIPCHK(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); });
        pres->prepare_device();
        // a horizontally uniform roughness length: the per-level mixing-length table (same bits as the per-cell evaluation)
        TF* mlen2 = nullptr;
        bool z0_uniform = true; for (TF v : z0m_h) z0_uniform = z0_uniform && (v == z0m_h[0]);
        if (z0_uniform)
        {
            HIPCHK(hipMalloc((void**)&mlen2, nk*sizeof(TF)));
            diff->set_uniform_z0m(z0m_h[0], false, mlen2, [](void* d, const void* s, size_t n) { HIPCHK(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); });
        }

        // ---- the slice of Model::exec this package accelerates (src/model.cxx:346-411) ----
        double cfl, dnum, div;
        if (!slab)
        {
            for (auto& it : fields.mp) boundary_cyclic.exec_g(it.second->fld_g);
            for (auto& it : fields.sp) boundary_cyclic.exec_g(it.second->fld_g);
            diff->exec_viscosity(thermo);
            cfl = advec->get_cfl(dt);
            dnum = diff->get_dn(dt);
            if (fused) diff->exec_with_advec(*advec, stats, nullptr, buoy ? &thermo : nullptr);
            else     { if (buoy) thermo.exec(grid, fields); advec->exec(stats); diff->exec(stats); }
            pres->exec(dt, stats);
            div = pres->check_divergence();
        }
        else
        {
            // the slab code path of a y-decomposed run (mhh_host_rccl.h) on a ONE-rank RCCL communicator: the north-south halos,
            // both transposes of the pressure solve and the scalar maxima go through ncclSend / ncclRecv / ncclAllReduce to self
            Master_rccl master;
            master.init(1, 0, Master_rccl::unique_id(), nullptr);
            Boundary_cyclic_slab<TF> halo(master, grid);
            Pres_slab<TF> pres_slab(master, grid, fields);
            pres_slab.set_reduce_workspace(work);
            pres_slab.prepare_device();
            if (sliced) { pres_slab.set_chunks(4); if (pres_slab.chunks() != 4) { std::fprintf(stderr, "set_chunks\n"); return 7; } }   // k-sliced transposes on a second stream
            if (overlap)
            {
                // the prognostic halos travel on a stream of their own while the rows that need none are worked; edge strips in one launch each
                Substep_slab<TF> sub(master, grid, fields, halo);
                if (!sub.can_overlap(*advec, *diff)) { std::fprintf(stderr, "can_overlap\n"); return 7; }
                sub.halo_visc_rhs(*advec, *diff, thermo);
                cfl = master.max(advec->get_cfl(dt));
                dnum = master.max(diff->get_dn(dt));
            }
            else
            {
            halo.exec_g({fields.mp["u"]->fld_g, fields.mp["v"]->fld_g, fields.mp["w"]->fld_g, fields.sp["th"]->fld_g});   // one message pair for all four
            diff->exec_viscosity(thermo);
            cfl = master.max(advec->get_cfl(dt));
            dnum = master.max(diff->get_dn(dt));
            if (fused) diff->exec_with_advec(*advec, stats, nullptr, buoy ? &thermo : nullptr);
            else     { if (buoy) thermo.exec(grid, fields); advec->exec(stats); diff->exec(stats); }
            }
            pres_slab.exec(dt, stats);
            div = pres_slab.check_divergence();
            HIPCHK(hipDeviceSynchronize());
            pres_slab.clear_device();
        }
        HIPCHK(hipDeviceSynchronize());

        FILE* out = std::fopen(argv[2], "wb");
        double sc[3] = {cfl, dnum, div}; std::fwrite(sc, sizeof(double), 3, out);
        dn(out, fields.mt["u"]->fld_g, n3); dn(out, fields.mt["v"]->fld_g, n3); dn(out, fields.mt["w"]->fld_g, n3); dn(out, fields.st["th"]->fld_g, n3);
        dn(out, fields.sd["evisc"]->fld_g, n3); dn(out, fields.sd["p"]->fld_g, n3);
        std::fclose(out);
        // error path: an illegal switch throws like the reference's factories
        bool threw = false;
        try { Advec<TF>::factory(grid, fields, "2i7"); } catch (const std::runtime_error&) { threw = true; }
        if (!threw) { std::fprintf(stderr, "factory did not throw\n"); return 4; }
        // the disabled operators (swadvec / swdiff / swpres = "0") exist and do nothing
        auto pres0 = Pres<TF>::factory(grid, fields, "0");
        pres0->prepare_device(); pres0->exec(dt, stats);
        if (pres0->check_divergence() != TF(0)) { std::fprintf(stderr, "Pres_disabled::check_divergence\n"); return 6; }
        pres->clear_device();
        std::printf("host_step ok cfl=%.17g dn=%.17g div=%.17g\n", cfl, dnum, div);
    }
    catch (const std::exception& e) { std::cerr << "EXCEPTION: " << e.what() << std::endl; return 5; }
    return 0;
}
