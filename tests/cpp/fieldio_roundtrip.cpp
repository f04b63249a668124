// tests/cpp/fieldio_roundtrip.cpp -- host-only check of the Field3d_io mirror: load a field file written by
// microhh_amd/fieldio.py (the reference's layout), save it again (optionally as two y-slabs), byte-compare in the test.
#include <cstdlib>
#include <iostream>
#include "../../microhh_amd/host/mhh_host.h"
using namespace mhh_host;

int main(int argc, char** argv)
{
    if (argc < 9) { std::cerr << "usage: fieldio_roundtrip in out itot jtot ktot igc jgc kgc [npy]\n"; return 1; }
    const int itot = std::atoi(argv[3]), jtot = std::atoi(argv[4]), ktot = std::atoi(argv[5]);
    const int igc = std::atoi(argv[6]), jgc = std::atoi(argv[7]), kgc = std::atoi(argv[8]);
    const int npy = argc > 9 ? std::atoi(argv[9]) : 1;
    int nerror = 0;
    for (int r=0; r<npy; ++r)
    {
        Grid<double> grid; auto& gd = grid.gd;
        gd.itot = itot; gd.jtot = jtot; gd.ktot = ktot; gd.igc = igc; gd.jgc = jgc; gd.kgc = kgc;
        gd.imax = itot; gd.jmax = jtot/npy; gd.kmax = ktot; gd.npy = npy; gd.mpicoordy = r;
        gd.icells = gd.imax + 2*igc; gd.jcells = gd.jmax + 2*jgc; gd.kcells = ktot + 2*kgc; gd.ijcells = gd.icells*gd.jcells; gd.ncells = gd.ijcells*gd.kcells;
        gd.istart = igc; gd.jstart = jgc; gd.kstart = kgc; gd.iend = igc + gd.imax; gd.jend = jgc + gd.jmax; gd.kend = kgc + ktot;
        std::vector<double> data(gd.ncells, -7.), tmp1((size_t)gd.imax*gd.jmax*gd.kmax), tmp2(1);
        Field3d_io<double> io(grid);
        nerror += io.load_field3d(data.data(), tmp1.data(), tmp2.data(), argv[1], 300., gd.kstart, gd.kend);
        nerror += io.save_field3d(data.data(), tmp1.data(), tmp2.data(), argv[2], 300., gd.kstart, gd.kend);
        if (data[0] != -7.) ++nerror;          // ghost cells are not touched by the load
    }
    std::cout << "nerror " << nerror << std::endl;
    return nerror;
}
