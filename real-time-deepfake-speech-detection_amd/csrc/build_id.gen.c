const char afx_build_id_str[] = "9b1572c5ee51";
