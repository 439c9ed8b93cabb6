#include <vector>
#include <random>
#include <chrono>
#include <cstdio>
#include <algorithm>
namespace svdf { void host_parallel_sort_scores(const float*, int*, long, int); }
int main(){ long n=100000; std::mt19937 g(1); std::normal_distribution<float> d; std::vector<float> sc(n); for(auto&x:sc)x=d(g);
 for(int th: {1,2,4,8,16,32}){ double best=1e9; for(int r=0;r<5;r++){ std::vector<int> ids(n); for(long i=0;i<n;i++)ids[i]=i; auto t0=std::chrono::steady_clock::now(); svdf::host_parallel_sort_scores(sc.data(),ids.data(),n,th); double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count(); best=std::min(best,ms);} printf("threads %d: %.2f ms\n",th,best);}
 struct E{int i; float s; bool operator<(const E&p)const{return s>p.s;}}; std::vector<E> e(n); for(long i=0;i<n;i++)e[i]={int(i),sc[i]}; auto t0=std::chrono::steady_clock::now(); std::sort(e.begin(),e.end()); printf("std::sort struct: %.2f ms\n", std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count()); }
