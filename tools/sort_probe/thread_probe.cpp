#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
int main(){ for(int T:{1,2,4,8}){ auto t0=std::chrono::steady_clock::now(); std::vector<std::thread> p; for(int t=0;t<T;t++)p.emplace_back([]{ volatile double x=0; for(long i=0;i<200000000;i++)x+=i;}); for(auto&t:p)t.join(); printf("%d threads: %.0f ms\n",T,std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count());}}
